"""Writes inputs.json for gen_compiler_golden.js: hand-picked regexes covering the dialect of src/vrm/regex.js:236-367
(literals, groups, | * + ?, backslash escapes, the epsilon character, digits as symbols — JS orders integer-like object
keys first, which reaches Hopcroft's symbol order) plus seeded random ones, and malformed inputs."""
import json
import random

hand = [
    "a", "ab", "a|b", "a*", "a+", "a?", "(a|b)*abb", "(a|b)+c?", "((a|b)+c)?d", "a(b|c)*d+", "(ab)*(ba)*", "(a|b|c)(a|b|c)(a|b|c)",
    "(0|1)*1(0|1)(0|1)", "(0|1|2|3|4|5|6|7|8|9)+", "x(0|1)+y(2|3)*z", "9a8b7c", "(a|0)(b|1)(c|2)*", "0", "10*1",
    "a\\*b", "\\(a\\)", "a\\|b", "\\\\", "\\n\\r\\t\\v\\f", "a\\nb|c\\td", "\\x0b", "a.b", "[ab]", "^a$", "a{2}", "\"q\"", "it's",
    "email was meant for @(a|b|c|d|e|f|g|h|i|j|k|l|m|n|o|p|q|r|s|t|u|v|w|x|y|z)+.",
    "(a|b|c|d|e|f|g|h|i|j|k|l|m|n|o|p|q|r|s|t|u|v|w|x|y|z)+@(a|b|c)+\\.(com|org)",
    "from:((a|b| )+<)?(a|b|_|\\.)+@(a|b|\\.)+>?\r\n", "(\r\n|\x0b|\x0c)+", "aϵb", "ϵ", "(a|ϵ)b", "(a*)*", "(a+)+", "(a?)?b",
    "((a))", "(((a|b)))*", "a**", "a+*", "a?+", "a|a", "(a|ab)(c|bcd)", "ab|abc|abcd", "(ab|a)*", "(a|b)*a(a|b)(a|b)(a|b)",
    "é", "café", "(é|ÿ)+x", "~}|{", " ", "\t", "a b", "A(B|C)*D", "(Z|Y|X)+(W|V)?U",
    # malformed
    "", "a|", "|a", "a||b", "()", "(a", "a)", "*a", "+", "?", "(*)", "a(|b)", "(a|)", "a)|b",
]   # a lone trailing backslash is left out: the JS reads past the end of the string (edge labelled "undefined")

rng = random.Random(20240917)
ATOMS = list("abc01") + ["\\n", "\\.", "\\*", "x", "Y", "7", " ", "\\\\", "-"]


def gen(depth):
    r = rng.random()
    if depth <= 0 or r < 0.30:
        return rng.choice(ATOMS)
    if r < 0.55:
        return "".join(gen(depth - 1) for _ in range(rng.randint(2, 4)))
    if r < 0.75:
        return "(" + "|".join(gen(depth - 1) for _ in range(rng.randint(2, 4))) + ")"
    if r < 0.85:
        return "(" + gen(depth - 1) + ")*"
    if r < 0.93:
        return "(" + gen(depth - 1) + ")+"
    return "(" + gen(depth - 1) + ")?"


rand = []
while len(rand) < 400:
    s = gen(rng.randint(2, 5))
    if len(s) <= 120:
        rand.append(s)
# longer ones: more subset states before minimisation, deeper Hopcroft queues
while len(rand) < 600:
    s = "".join(gen(rng.randint(3, 6)) for _ in range(rng.randint(2, 5)))
    if 40 <= len(s) <= 400:
        rand.append(s)
# digit-heavy alphabets (integer-like object keys) and prefix-sharing alternations
DIG = list("0123456789ab")
for _ in range(100):
    words = ["".join(rng.choice(DIG) for _ in range(rng.randint(1, 5))) for _ in range(rng.randint(2, 7))]
    tail = rng.choice(["", "*", "+", "?"])
    rand.append("(" + "|".join(words) + ")" + tail + rng.choice(["", "x", "(0|1)*", "9+"]))
json.dump({"small": hand + rand, "big": ["regex1_test", "regex2_test", "regex3_test", "ex_regex", "header_from", "header_to", "header_subject"]}, open("inputs.json", "w"), indent=0,
          ensure_ascii=True)
print(len(hand), len(rand))
