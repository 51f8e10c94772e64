"""Integer MockProver for the witness rows (test infrastructure only; never imported by the product).

The reference's strongest test is `MockProver::verify()` (src/lib.rs:1086,1113,1336,1364, ...): it accepts a witness iff
every gate and lookup of `RegexVerifyConfig::configure` (src/lib.rs:126-305) holds on the assigned cells.  This module
restates those constraints over plain integers and evaluates them on the compact witness rows a kernel wrote:

  gate  "The state must start from the first state value"   src/lib.rs:173-191   -> FAIL_FIRST_STATE
  gate  "The transition of enable flags"                    src/lib.rs:193-205   -> FAIL_ENABLE
  lookup "characters and their state"                       src/lib.rs:207-233   -> FAIL_TRANSITION
         (enable*char, enable*cur + !enable*dummy, enable*next + !enable*dummy, enable*substr_id) in transition rows
  lookup "start_state of substring"                         src/lib.rs:235-259   -> FAIL_START_LOOKUP
         (se*substr_id, se*cur + !se*dummy, dummy) in endpoint rows
  lookup "end_state of substring"                           src/lib.rs:261-284   -> FAIL_END_LOOKUP
         (ee*substr_id, dummy, ee*next + !ee*dummy) in endpoint rows
  accept chain  assert_equal(select(is_state_eq, 1, pre_enable - cur_enable), 1)   src/lib.rs:427-457 -> FAIL_ACCEPT
         (halo2-base select(a, b, sel) = sel ? a : b: where enable drops from 1 to 0 the state must be the accepted one)

The fixed tables are the rows `RegexTableConfig::load` assigns (src/table.rs:61-198), taken from the library's
hrx_table_transition_rows / hrx_table_endpoint_rows (which tests/test_abi.py and tests/test_oracle_golden.py pin to the
reference's fixture files) — NOT from any walk.  So a witness that passes is pinned to the reference's constraint system
independently of the oracle's (and the kernels') DFA walk: the lookups force every enabled row's (char, state, next state,
substr id) into the table, which for a deterministic DFA fixes the whole state / substr-id trace from the first state.

verify() accepts witnesses whose start_enable / end_enable bits are 0 where the walk would set them (the lookups only
check the bits that are set).  Two further checks, outside what verify() sees, pin those and the rest of the assignment:

  FAIL_FLAGS     start_enable / end_enable are exactly the endpoint-table memberships derive_is_start_end computes
                 (src/lib.rs:847-888) gated by enable (src/lib.rs:482-519; row M-1 has no end_enable cell)
  FAIL_PADDING   rows at and after the string's end hold what src/lib.rs:387-418 assigns (substr id 0, flags 0, the
                 dummy state after row n)
  FAIL_MASK      masked_char / masked_substr_id equal the gate chain of src/lib.rs:593-764 evaluated on the RECORDS'
                 own Sum(substr_id), Sum(is_start), Sum(is_end): forward "last set/reset event wins" scan, the same
                 backward, mask = start_mask * end_mask

Everything is written with torch tensor operations (index lookups, cummax scans), so a full-size device-resident batch is
checked where it lies; the same code runs on CPU tensors in the non-GPU tests.  One string = one row of the result.
"""
import numpy as np
import torch

FAIL_FIRST_STATE, FAIL_ENABLE, FAIL_TRANSITION, FAIL_START_LOOKUP, FAIL_END_LOOKUP, FAIL_ACCEPT = 1, 2, 4, 8, 16, 32
FAIL_FLAGS, FAIL_PADDING, FAIL_MASK = 64, 128, 256
VERIFY_BITS = FAIL_FIRST_STATE | FAIL_ENABLE | FAIL_TRANSITION | FAIL_START_LOOKUP | FAIL_END_LOOKUP | FAIL_ACCEPT
NAMES = {FAIL_FIRST_STATE: "first-state gate", FAIL_ENABLE: "enable gate", FAIL_TRANSITION: "transition lookup",
         FAIL_START_LOOKUP: "start lookup", FAIL_END_LOOKUP: "end lookup", FAIL_ACCEPT: "accept chain",
         FAIL_FLAGS: "flag completeness", FAIL_PADDING: "padding", FAIL_MASK: "masked rows"}


class _DefTables:
    """Dense membership arrays of one def's fixed tables (src/table.rs:61-198), on `device`."""

    def __init__(self, transition_rows, endpoint_rows, first_state, accepted_state, device):
        tr = np.asarray(transition_rows, dtype=np.int64)       # (char, cur, next, substr_id), row 0 = (0, dummy, dummy, 0)
        er = np.asarray(endpoint_rows, dtype=np.int64)         # (substr_id, start, end),      row 0 = (0, dummy, dummy)
        assert tr[0, 0] == 0 and tr[0, 1] == tr[0, 2] and tr[0, 3] == 0, "table.rs:98: the first transition row is the dummy row"
        self.dummy = int(tr[0, 1])
        assert er[0, 0] == 0 and er[0, 1] == self.dummy and er[0, 2] == self.dummy, "table.rs:126-144: the first endpoint row is the dummy row"
        self.first, self.accepted = int(first_state), int(accepted_state)
        self.ns = self.dummy + 1                                # states 0 .. dummy
        key = tr[:, 1] * 256 + tr[:, 0]
        # (char as u8, cur) keys are unique: defs.rs:100 stores them in a HashMap
        assert len(np.unique(key)) == len(key), "duplicate (char, cur_state) rows in the transition table"
        t_next = np.full(self.ns * 256, -1, np.int64)
        t_sid = np.full(self.ns * 256, -1, np.int64)
        t_next[key], t_sid[key] = tr[:, 2], tr[:, 3]
        self.t_next, self.t_sid = torch.from_numpy(t_next).to(device), torch.from_numpy(t_sid).to(device)
        # endpoint rows with end == dummy serve the start lookup (third component is the constant dummy), rows with
        # start == dummy the end lookup; the dummy row serves both
        self.nsid = int(er[:, 0].max()) + 1
        sm = np.zeros(self.nsid * self.ns, bool)
        em = np.zeros(self.nsid * self.ns, bool)
        s_rows, e_rows = er[er[:, 2] == self.dummy], er[er[:, 1] == self.dummy]
        sm[s_rows[:, 0] * self.ns + s_rows[:, 1]] = True
        em[e_rows[:, 0] * self.ns + e_rows[:, 2]] = True
        self.start_member, self.end_member = torch.from_numpy(sm).to(device), torch.from_numpy(em).to(device)


class IntegerMockProver:
    def __init__(self, tables, device="cpu"):
        """tables: per def (transition_rows, endpoint_rows, first_state, accepted_state)"""
        self.device = torch.device(device)
        self.defs = [_DefTables(*t, device=self.device) for t in tables]
        self.D = len(self.defs)

    @classmethod
    def from_config(cls, cfg, device="cpu"):
        """cfg: halo2_regex_amd.RegexVerifyConfig: load() returns the rows RegexTableConfig::load assigns, in table.rs order;
        first_state / accepted_state are AllstrRegexDef's header lines (defs.rs:84-99)"""
        rows = cfg.load()
        return cls([(rows[d][0], rows[d][1], cfg.first_state(d), cfg.accepted_state(d)) for d in range(len(rows))], device)

    @torch.no_grad()
    def verify(self, chars, lens, records, masked, M):
        """chars (B, stride) u8, lens (B,), records (B, >= M, D) 32-bit, masked (B, >= M) 16-bit (string-major views; any
        device).  Returns a (B,) int64 tensor of FAIL_* bits, 0 = every constraint and assignment rule holds."""
        dev = self.device
        B = int(lens.shape[0])
        lens = torch.as_tensor(lens).to(dev).to(torch.int64)
        rows = torch.arange(M, device=dev)
        en = rows[None, :] < lens[:, None]                                     # lib.rs:339-348
        ch = torch.zeros((B, M), dtype=torch.int64, device=dev)
        w = min(M, chars.shape[1])
        ch[:, :w] = chars[:, :w].to(dev).to(torch.int64)
        ch = ch * en
        fail = torch.zeros(B, dtype=torch.int64, device=dev)
        in_contract = lens <= M

        def flag(cond_rows, bit):                                              # cond_rows: (B, M) or (B,) bool, True = violated
            nonlocal fail
            bad = cond_rows.any(dim=1) if cond_rows.dim() == 2 else cond_rows
            fail |= bad.to(torch.int64) * bit

        # gate "The transition of enable flags" (lib.rs:193-205): enable is boolean and never rises — true of [r < n] by
        # construction; evaluated anyway so that the checker mirrors verify() gate for gate
        e64 = en.to(torch.int64)
        chg = e64[:, :-1] - e64[:, 1:]
        flag((chg * (1 - chg)) != 0, FAIL_ENABLE)

        rec = records[:, :M].to(dev).to(torch.int64) & 0xffffffff
        SID = torch.zeros((B, M), dtype=torch.int64, device=dev)
        ST = torch.zeros((B, M + 1), dtype=torch.int64, device=dev)            # lib.rs:380-385: M + 1 cells, the last stays 0
        EN = torch.zeros((B, M + 1), dtype=torch.int64, device=dev)
        for d, T in enumerate(self.defs):
            r = rec[:, :, d]
            state, sid, fl = r & 0xffff, (r >> 16) & 0xff, r >> 24
            se, ee = fl & 1, (fl >> 1) & 1
            flag(fl > 3, FAIL_PADDING)
            # Rotation::next of the last row is a cell match_substrs never assigns (App. A.2: callers keep n < M); with
            # enable[M-1] = 0 it does not matter, with n == M that one lookup is left out (marked below)
            nxt = torch.cat([state[:, 1:], torch.full((B, 1), T.dummy, dtype=torch.int64, device=dev)], dim=1)
            last_unassigned = torch.zeros((B, M), dtype=torch.bool, device=dev)
            last_unassigned[:, M - 1] = en[:, M - 1]
            # ---- gate "The state must start from the first state value" (lib.rs:173-191), row 0 only
            flag(en[:, 0] & (state[:, 0] != T.first), FAIL_FIRST_STATE)
            # ---- lookup "characters and their state" (lib.rs:207-233)
            cur_t = torch.where(en, state, torch.full_like(state, T.dummy))
            nxt_t = torch.where(en, nxt, torch.full_like(state, T.dummy))
            sid_t = sid * e64
            in_range = cur_t < T.ns
            key = torch.where(in_range, cur_t, torch.zeros_like(cur_t)) * 256 + ch
            ok = in_range & (T.t_next[key] == nxt_t) & (T.t_sid[key] == sid_t)
            flag(~ok & ~last_unassigned, FAIL_TRANSITION)
            # ---- lookup "start_state of substring" (lib.rs:235-259): (se*sid, se*cur + (1-se)*dummy, dummy)
            s_sid = se * sid
            s_st = torch.where(se == 1, state, torch.full_like(state, T.dummy))
            okr = (s_sid < T.nsid) & (s_st < T.ns)
            k = torch.where(okr, s_sid * T.ns + s_st, torch.zeros_like(s_sid))
            flag(~(okr & T.start_member[k]), FAIL_START_LOOKUP)
            # ---- lookup "end_state of substring" (lib.rs:261-284): (ee*sid, dummy, ee*next + (1-ee)*dummy)
            e_sid = ee * sid
            e_st = torch.where(ee == 1, nxt, torch.full_like(state, T.dummy))
            okr = (e_sid < T.nsid) & (e_st < T.ns)
            k = torch.where(okr, e_sid * T.ns + e_st, torch.zeros_like(e_sid))
            flag(~(okr & T.end_member[k]) & ~last_unassigned, FAIL_END_LOOKUP)
            # ---- accept chain (lib.rs:427-457): rows 0 .. M-1; pre = 1 at row 0, else enable[r-1]; where pre - cur = 1
            # the state must be the accepted one
            pre = torch.cat([torch.ones((B, 1), dtype=torch.int64, device=dev), e64[:, :-1]], dim=1)
            drop = (pre - e64) == 1
            flag(drop & (state != T.accepted), FAIL_ACCEPT)
            # ---- beyond verify(): the flags are exactly derive_is_start_end's (lib.rs:847-888) gated by enable (lib.rs:482-519)
            tagged = en & (sid != 0)
            sid_c = torch.where(sid < T.nsid, sid, torch.zeros_like(sid))
            st_c = torch.where(state < T.ns, state, torch.zeros_like(state))
            nx_c = torch.where(nxt < T.ns, nxt, torch.zeros_like(nxt))
            exp_se = tagged & T.start_member[sid_c * T.ns + st_c]
            exp_ee = tagged & T.end_member[sid_c * T.ns + nx_c]
            exp_ee[:, M - 1] = False                                           # lib.rs:501: idx runs to M - 2
            flag((exp_se.to(torch.int64) != se) | (exp_ee.to(torch.int64) != ee), FAIL_FLAGS)
            # ---- beyond verify(): padding (lib.rs:387-418): substr id 0 from row n on, the dummy state after row n
            after = rows[None, :] > lens[:, None]
            flag((~en & ((sid != 0) | (fl != 0))) | (after & (state != T.dummy)), FAIL_PADDING)
            SID += sid * e64                                                   # lib.rs:467-471
            ST[:, :M] += se                                                    # lib.rs:494-498 (rows < n are enabled, the rest is 0)
            EN[:, 1:M] += ee[:, :M - 1]                                        # lib.rs:501-519: EN[r + 1] for r = 0 .. M - 2
        # ---- reveal masks (lib.rs:593-764) from the records' own sums
        zero = torch.zeros((B, 1), dtype=torch.int64, device=dev)

        def last_event_scan(is_set, is_reset):
            """new = reset ? 0 : (set ? 1 : previous), previous = 0 before the first row: the last row with an event decides"""
            ev = torch.where(is_reset, 2, torch.where(is_set, 1, 0))
            idx = torch.where(ev > 0, rows[None, :].expand(B, M), torch.full((B, M), -1, dtype=torch.int64, device=dev))
            last = torch.cummax(idx, dim=1).values
            return (last >= 0) & (torch.gather(ev, 1, last.clamp(min=0)) == 1)

        chg_f = torch.cat([zero, SID[:, :-1]], dim=1) != SID                  # pre_substr_id = 0 at row 0
        st, enr = ST[:, :M] != 0, EN[:, :M] != 0
        overlap = ((ST > 1) | (EN > 1)).any(dim=1)                             # two defs flag one row: out of contract (App. A.3)
        sm = last_event_scan(st & chg_f, ~st & enr & chg_f)
        # backward: position p = M - 1 - idx; set = EN[p + 1] & changed, reset = !EN[p + 1] & ST[p + 1] & changed, changed = SID[p + 1] != SID[p] (SID[M] = 0)
        chg_b = torch.cat([SID[:, 1:], zero], dim=1) != SID
        en1, st1 = EN[:, 1:] != 0, ST[:, 1:] != 0
        em = last_event_scan((en1 & chg_b).flip(1), (~en1 & st1 & chg_b).flip(1)).flip(1)
        mask = (sm & em).to(torch.int64)
        exp = (mask * ch) | ((mask * SID) << 8)
        got = masked[:, :M].to(dev).to(torch.int64) & 0xffff
        flag((exp != got) & ~overlap[:, None], FAIL_MASK)
        fail = torch.where(in_contract, fail, torch.full_like(fail, -1))
        return fail

    @staticmethod
    def explain(code):
        code = int(code)
        return "ok" if code == 0 else " + ".join(v for k, v in NAMES.items() if code & k)
