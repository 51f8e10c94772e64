// Reads like src/lib.rs:1067-1092 (test_substr_pass1), through the C++ mirror of the reference surface.
// argv[1] = directory with the DFA fixtures; argv[2] = "gpu" to run the compute part (needs an MI355X).
#include <cstdio>
#include <cstring>
#include <string>

#include "../../halo2_regex_amd/csrc/hrx_host.hpp"
using namespace hrx_host;

int main(int argc, char **argv) {
    const std::string dir = argc > 1 ? argv[1] : "tests/golden/dfa";
    const bool gpu = argc > 2 && !strcmp(argv[2], "gpu");
    const size_t MAX_STRING_LEN = 1024;  // lib.rs:930
    std::vector<RegexDefs> regex_defs = {
        {AllstrRegexDef::read_from_text(dir + "/regex1_test_lookup.txt"), {SubstrRegexDef::read_from_text(dir + "/substr1_test_lookup.txt")}},
        {AllstrRegexDef::read_from_text(dir + "/regex2_test_lookup.txt"), {SubstrRegexDef::read_from_text(dir + "/substr2_test_lookup.txt")}},
    };  // lib.rs:978-987
    RegexVerifyConfig config = RegexVerifyConfig::configure(MAX_STRING_LEN, regex_defs, gpu ? 0 : -1);
    auto rows = config.load_transition_rows();
    if (rows[0].size() != 4 * 2843 || rows[1].size() != 4 * 1275) { printf("table rows wrong\n"); return 1; }
    if (!gpu) {
        bool threw = false;
        try { RegexVerifyConfig::configure(8, {{AllstrRegexDef::from_string("0\n1\n1\n0 x 97\n"), {}}}, -1); } catch (const std::runtime_error &) { threw = true; }
        printf(threw ? "host ok\n" : "parse error not raised\n");
        return threw ? 0 : 1;
    }
    const std::string s = "email was meant for @y. Also for x.";
    const std::vector<uint8_t> characters(s.begin(), s.end());
    const AssignedRegexResult result = config.match_substrs(characters);
    std::vector<uint64_t> expected_masked_chars(MAX_STRING_LEN, 0), expected_substr_ids(MAX_STRING_LEN, 0);
    const std::pair<size_t, std::string> correct_substrs[] = {{21, "y"}, {33, "x"}};  // lib.rs:1081
    size_t substr_idx = 0;
    for (auto &cs : correct_substrs) {
        for (size_t idx = 0; idx < cs.second.size(); ++idx) {
            expected_masked_chars[cs.first + idx] = (uint8_t)cs.second[idx];
            expected_substr_ids[cs.first + idx] = substr_idx + 1;
        }
        ++substr_idx;
    }
    if (result.masked_characters != expected_masked_chars || result.all_substr_ids != expected_substr_ids) { printf("mismatch\n"); return 1; }
    auto states = config.derive_states(characters);
    if (states[0][22] != 22 || states[0][23] != 24 || states[1].back() != 12) { printf("states wrong\n"); return 1; }
    bool threw = false;
    try { config.derive_states({200}); } catch (const std::runtime_error &e) { threw = std::string(e.what()) == "The transition from 0 by 200 is invalid!"; }
    printf(threw ? "gpu ok\n" : "panic text wrong\n");
    return threw ? 0 : 1;
}
