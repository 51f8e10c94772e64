// Generates tests/golden/compiler/cases.json: regex -> (DFA JSON of regexToDfa, allstr definition text) by RUNNING the
// reference's src/vrm/regex.js under node (in this container only; /root/reference does not travel) and applying the
// ordering of dfa_to_regex_def_text (src/vrm/js_caller.rs:127-157; serde_json's Map is a BTreeMap, so edge keys come
// out in byte order of the key text).  Usage:  node gen_compiler_golden.js  (from this directory)
// Inputs: inputs.json (small regexes, written by hand / by gen_inputs.py) and the reference's own decomposed-regex
// JSON fixtures (tests/golden/dfa/regex{1,2,3}_test.json, copied data files).
const fs = require("fs"), vm = require("vm"), path = require("path"), crypto = require("crypto");
const ctx = vm.createContext({ console: console });
vm.runInContext(fs.readFileSync("/root/reference/src/vrm/regex.js", "utf8"), ctx);

function cmpBytes(a, b) { return Buffer.compare(Buffer.from(a, "utf8"), Buffer.from(b, "utf8")); }

function allstrText(graph) {
    let accepted = -1, maxState = 0;
    graph.forEach((n, i) => { if (accepted < 0 && n.type === "accept") accepted = i; });
    if (accepted < 0) return "ERR NoAcceptedState";
    graph.forEach(n => Object.values(n.edges).forEach(v => { if (v > maxState) maxState = v; }));
    let text = "0\n" + accepted + "\n" + maxState + "\n";
    graph.forEach((n, i) => {
        Object.keys(n.edges).sort(cmpBytes).forEach(key => {
            JSON.parse(key).forEach(ch => {
                text += i + " " + n.edges[key] + " " + (ch.codePointAt(0) & 0xff) + "\n";
            });
        });
    });
    return text;
}

// the parser's own message for a malformed pattern: parseRegex returns it as a string ("Error: unexpected * at 0.", regex.js:236-367); regexToDfa then throws on it
function errorOf(regex) {
    let text = null;
    try { const p = ctx.parseRegex(regex); if (typeof p === "string") text = p; } catch (e) { text = null; }
    return text === null ? { regex: regex, error: true } : { regex: regex, error: true, error_text: text };
}

function run(regex, full) {
    let json;
    try { json = ctx.regexToDfa(regex); } catch (e) { return errorOf(regex); }
    let graph;
    try { graph = JSON.parse(json); } catch (e) { return errorOf(regex); }
    if (!Array.isArray(graph) || graph.some(n => n === null)) return errorOf(regex);
    const text = allstrText(graph), sha = t => crypto.createHash("sha256").update(t, "utf8").digest("hex");
    if (!full && json.length + text.length > 1500)   // keep the fixture small: big results are pinned by their digests
        return { regex: regex, states: graph.length, dfa_json_sha256: sha(json), allstr_sha256: sha(text) };
    return { regex: regex, states: graph.length, dfa_json: json, allstr: text };
}

const inputs = JSON.parse(fs.readFileSync(path.join(__dirname, "inputs.json"), "utf8"));
const out = { small: [], big: [] };
for (const r of inputs.small) out.small.push(run(r));
for (const name of inputs.big) {
    const cfg = JSON.parse(fs.readFileSync(path.join(__dirname, "..", "dfa", name + ".json"), "utf8"));
    const regex = cfg.parts.map(p => p.regex_def).join("");
    const r = run(regex, true);
    out.big.push({ name: name, regex: regex, dfa_json_sha256: crypto.createHash("sha256").update(r.dfa_json, "utf8").digest("hex"),
                   allstr_sha256: crypto.createHash("sha256").update(r.allstr, "utf8").digest("hex"),
                   allstr_file: name === "ex_regex" ? "ex_allstr.txt" : name + "_lookup.txt" });
}
fs.writeFileSync(path.join(__dirname, "cases.json"), JSON.stringify(out, null, 1));
console.log("small", out.small.length, "errors", out.small.filter(c => c.error).length, "big", out.big.length);
