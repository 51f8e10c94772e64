"""regex -> minimal DFA -> AllstrRegexDef text against vectors produced by RUNNING the reference's
src/vrm/regex.js under node (tests/golden/compiler/gen_compiler_golden.js) and against the reference's committed
definition files.  Host-only code: no GPU needed."""
import hashlib
import json
import os

import pytest

import halo2_regex_amd as hra

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "compiler", "cases.json")))
DFA_DIR = os.path.join(HERE, "golden", "dfa")


def test_small_regexes_json_and_text_are_byte_identical():
    n_ok = n_err = biggest = 0
    for c in CASES["small"]:
        if c.get("error"):
            with pytest.raises(hra.HrxError) as e:
                hra.regex_to_allstr_text(c["regex"])
            assert e.value.code == hra.HRX_ERR_PARSE and str(e.value).startswith("Error:")
            assert str(e.value) == c["error_text"], c["regex"]     # the parser's own message, text and position (regex.js:236-367)
            n_err += 1
            continue
        js, text = hra.regex_to_dfa_json_text(c["regex"]), hra.regex_to_allstr_text(c["regex"])
        if "dfa_json" in c:
            assert js == c["dfa_json"] and text == c["allstr"], c["regex"]
        else:
            assert hashlib.sha256(js.encode("utf-8")).hexdigest() == c["dfa_json_sha256"], c["regex"]
            assert hashlib.sha256(text.encode("utf-8")).hexdigest() == c["allstr_sha256"], c["regex"]
        assert len(json.loads(js)) == c["states"]
        n_ok += 1
        biggest = max(biggest, c["states"])
    assert n_ok > 700 and n_err >= 10 and biggest >= 20


@pytest.mark.parametrize("case", CASES["big"], ids=[c["name"] for c in CASES["big"]])
def test_reference_definition_files_are_reproduced(case):
    """regex{1,2,3}_test.json / the example's parts -> exactly the committed *_lookup.txt / ex_allstr.txt"""
    cfg = hra.DecomposedRegexConfig.from_json(open(os.path.join(DFA_DIR, case["name"] + ".json")).read())
    assert cfg.all_regex() == case["regex"]
    text = cfg.gen_allstr_text()
    assert text == open(os.path.join(DFA_DIR, case["allstr_file"])).read()
    assert hashlib.sha256(text.encode()).hexdigest() == case["allstr_sha256"]
    js = hra.regex_to_dfa_json_text(case["regex"])
    assert hashlib.sha256(js.encode("utf-8")).hexdigest() == case["dfa_json_sha256"]


def test_compiled_definition_feeds_the_defs_parser():
    text = hra.regex_to_allstr_text("(a|b)*abb")
    lines = text.split("\n")
    assert lines[0] == "0" and all(len(l.split()) == 3 for l in lines[3:] if l)
    v = hra.get_dfa_json_value("(a|b)*abb")
    assert [n["type"] for n in v].count("accept") == 1 and len(v) == 4


def test_lone_trailing_backslash_is_rejected():
    with pytest.raises(hra.HrxError):
        hra.regex_to_allstr_text("ab\\")


def test_deep_nesting_is_a_parse_error_not_a_crash():
    """the reference's JS throws a catchable error when its stack runs out; the library returns HRX_ERR_PARSE (include/hrx.h:
    integer status returns, nothing unwinds) instead of overflowing the native stack"""
    assert hra.regex_to_allstr_text("(" * 1500 + "a" + ")" * 1500).startswith("0\n")
    for n in (2500, 200000):
        with pytest.raises(hra.HrxError) as e:
            hra.regex_to_allstr_text("(" * n + "a" + ")" * n)
        assert e.value.code == hra.HRX_ERR_PARSE and "nested deeper" in str(e.value)
