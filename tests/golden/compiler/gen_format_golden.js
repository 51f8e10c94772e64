// Generates format_cases.json: regex_def -> formatRegexPrintable(regex_def) by RUNNING the reference's
// src/vrm/regex.js:24-39 under node.  Node 12 lacks String.prototype.replaceAll; the polyfill below is the spec's
// behaviour for a string pattern (global replace of the escaped literal, `$` patterns in the replacement honoured).
const fs = require("fs"), vm = require("vm"), path = require("path");
const ctx = vm.createContext({ console: console });
const polyfill = [
    "if (!String.prototype.replaceAll) String.prototype.replaceAll = function (a, b) {",
    "    return this.replace(new RegExp(a.replace(/[.*+?^${}()|[\\]\\\\]/g, '\\\\$&'), 'g'), b); };",
].join("\n");
vm.runInContext(polyfill, ctx);
vm.runInContext(fs.readFileSync("/root/reference/src/vrm/regex.js", "utf8"), ctx);
const inputs = JSON.parse(fs.readFileSync(path.join(__dirname, "inputs.json"), "utf8"));
const extra = JSON.parse(fs.readFileSync(path.join(__dirname, "format_inputs.json"), "utf8"));
let defs = inputs.small.slice(0, 80).concat(extra);
for (const name of inputs.big) {
    const cfg = JSON.parse(fs.readFileSync(path.join(__dirname, "..", "dfa", name + ".json"), "utf8"));
    cfg.parts.forEach(p => defs.push(p.regex_def));
}
const out = defs.map(d => ({ regex_def: d, formatted: ctx.formatRegexPrintable(d) }));
fs.writeFileSync(path.join(__dirname, "format_cases.json"), JSON.stringify(out, null, 0));
console.log(out.length);
