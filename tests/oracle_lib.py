"""ctypes binding of the CPU oracle (oracle/hrx_oracle.c).  Test infrastructure only."""
import ctypes as C
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
DFA_DIR = os.path.join(GOLDEN, "dfa")

ORC_OK, ORC_INVALID_TRANSITION, ORC_FLAG_OVERLAP, ORC_BAD_LENGTH = 0, 1, 2, 3

_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def build_oracle():
    so = os.path.join(ROOT, "oracle", "_build", "libhrx_oracle.so")
    src = os.path.join(ROOT, "oracle", "hrx_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return so


def load_oracle():
    lib = C.CDLL(build_oracle())
    lib.orc_new.restype = C.c_void_p
    lib.orc_free.argtypes = [C.c_void_p]
    lib.orc_push_allstr_text.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.orc_push_substr_text.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    for f in ("orc_num_defs",):
        getattr(lib, f).argtypes = [C.c_void_p]
        getattr(lib, f).restype = C.c_size_t
    for f in ("orc_num_substrs", "orc_num_transitions"):
        getattr(lib, f).argtypes = [C.c_void_p, C.c_size_t]
        getattr(lib, f).restype = C.c_size_t
    for f in ("orc_first_state", "orc_accepted_state", "orc_largest_state"):
        getattr(lib, f).argtypes = [C.c_void_p, C.c_size_t]
        getattr(lib, f).restype = C.c_uint64
    lib.orc_derive_states.argtypes = [C.c_void_p, _u8p, C.c_size_t, _u64p, _u64p]
    lib.orc_derive_substr_ids.argtypes = [C.c_void_p, _u64p, C.c_size_t, _u64p]
    lib.orc_derive_is_start_end.argtypes = [C.c_void_p, _u64p, _u64p, C.c_size_t, _u8p, _u8p]
    lib.orc_match_substrs.argtypes = [C.c_void_p, _u8p, C.c_size_t, C.c_size_t] + [_u64p] * 9
    lib.orc_witness_batch.argtypes = [C.c_void_p, _u8p, C.c_size_t, C.POINTER(C.c_uint32), C.c_size_t, C.c_size_t,
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_uint16), _u64p]
    lib.orc_witness_batch_mt.argtypes = lib.orc_witness_batch.argtypes + [C.c_size_t]
    lib.orc_dense_new.argtypes = [C.c_void_p]
    lib.orc_dense_new.restype = C.c_void_p
    lib.orc_dense_free.argtypes = [C.c_void_p]
    lib.orc_dense_witness_batch.argtypes = lib.orc_witness_batch.argtypes + [C.c_size_t]
    lib.orc_table_transition_rows.argtypes = [C.c_void_p, C.c_size_t, _u64p, C.c_size_t]
    lib.orc_table_transition_rows.restype = C.c_size_t
    lib.orc_table_endpoint_rows.argtypes = [C.c_void_p, C.c_size_t, _u64p, C.c_size_t]
    lib.orc_fr_from_u64.argtypes = [C.c_uint64, _u64p]
    lib.orc_fr_from_u64.restype = None
    lib.orc_fr_constants.argtypes = [_u64p, _u64p, _u64p]
    lib.orc_fr_constants.restype = None
    lib.orc_table_endpoint_rows.restype = C.c_size_t
    return lib


class OracleDefs:
    """Vec<RegexDefs> held by the oracle.  defs = [(allstr_text, [substr_text, ...]), ...]"""

    def __init__(self, lib, defs):
        self.lib = lib
        self.h = lib.orc_new()
        for allstr, substrs in defs:
            a = allstr.encode() if isinstance(allstr, str) else allstr
            rc = lib.orc_push_allstr_text(self.h, a, len(a))
            if rc:
                raise ValueError("allstr parse error at line %d" % (-rc - 1))
            for s in substrs:
                s = s.encode() if isinstance(s, str) else s
                rc = lib.orc_push_substr_text(self.h, s, len(s))
                if rc:
                    raise ValueError("substr parse error at line %d" % (-rc - 1))
        self.D = lib.orc_num_defs(self.h)

    def __del__(self):
        try:
            if getattr(self, "_dense", None):
                self.lib.orc_dense_free(self._dense)
            self.lib.orc_free(self.h)
        except Exception:
            pass

    @classmethod
    def from_files(cls, lib, defs, base=DFA_DIR):
        out = []
        for allstr, substrs in defs:
            out.append((open(os.path.join(base, allstr), "rb").read(),
                        [open(os.path.join(base, s), "rb").read() for s in substrs]))
        return cls(lib, out)

    # --- src/lib.rs:804-888 -------------------------------------------------
    def derive_states(self, chars):
        chars = np.frombuffer(bytes(chars), dtype=np.uint8)
        n = len(chars)
        states = np.zeros((self.D, n + 1), dtype=np.uint64)
        info = np.zeros(5, dtype=np.uint64)
        rc = self.lib.orc_derive_states(self.h, _p(chars, _u8p), n, _p(states, _u64p), _p(info, _u64p))
        if rc:
            raise RuntimeError("The transition from %d by %d is invalid!" % (info[2], info[3]))
        return states

    def derive_substr_ids(self, states):
        n = states.shape[1] - 1
        sids = np.zeros((self.D, max(n, 0)), dtype=np.uint64)
        self.lib.orc_derive_substr_ids(self.h, _p(np.ascontiguousarray(states), _u64p), n, _p(sids, _u64p))
        return sids

    def derive_is_start_end(self, states, sids):
        n = states.shape[1] - 1
        st = np.zeros((self.D, n + 1), dtype=np.uint8)
        en = np.zeros((self.D, n + 1), dtype=np.uint8)
        self.lib.orc_derive_is_start_end(self.h, _p(np.ascontiguousarray(states), _u64p),
                                         _p(np.ascontiguousarray(sids), _u64p), n, _p(st, _u8p), _p(en, _u8p))
        return st.astype(bool), en.astype(bool)

    # --- src/lib.rs:311-773 (integers) --------------------------------------
    def match_substrs(self, chars, M):
        chars = np.frombuffer(bytes(chars), dtype=np.uint8)
        n = len(chars)
        D = self.D
        out = {
            "enable": np.zeros(M, np.uint64), "character": np.zeros(M, np.uint64),
            "state": np.zeros((D, M), np.uint64), "substr_id": np.zeros((D, M), np.uint64),
            "start_enable": np.zeros((D, M), np.uint64), "end_enable": np.zeros((D, M), np.uint64),
            "masked_char": np.zeros(M, np.uint64), "masked_substr_id": np.zeros(M, np.uint64),
        }
        info = np.zeros(5, np.uint64)
        rc = self.lib.orc_match_substrs(self.h, _p(chars, _u8p), n, M,
                                        *[_p(out[k], _u64p) for k in ("enable", "character", "state", "substr_id",
                                                                      "start_enable", "end_enable", "masked_char",
                                                                      "masked_substr_id")], _p(info, _u64p))
        out["rc"] = rc
        out["info"] = info
        return out

    # --- compact batch ------------------------------------
    def witness_batch(self, chars2d, lens, M, threads=1, dense=False, out=None):
        """chars2d: (B, stride) uint8; lens: (B,) uint32 -> records (B,M,D) u32, masked (B,M) u16, status (B,) u64.
        threads > 1: one string per task over host threads; dense=True: the dense-table "best CPU" variant."""
        chars2d = np.ascontiguousarray(chars2d, dtype=np.uint8)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        B, stride = chars2d.shape
        if out is None:
            out = np.zeros((B, M, self.D), np.uint32), np.zeros((B, M), np.uint16), np.zeros(B, np.uint64)
        rec, msk, status = out
        args = (_p(chars2d, _u8p), stride, _p(lens, C.POINTER(C.c_uint32)), B, M,
                _p(rec, C.POINTER(C.c_uint32)), _p(msk, C.POINTER(C.c_uint16)), _p(status, _u64p))
        if dense:
            if getattr(self, "_dense", None) is None:
                self._dense = self.lib.orc_dense_new(self.h)
            self.lib.orc_dense_witness_batch(self._dense, *args, threads)
        elif threads > 1:
            self.lib.orc_witness_batch_mt(self.h, *args, threads)
        else:
            self.lib.orc_witness_batch(self.h, *args)
        return rec, msk, status

    # --- src/table.rs:61-198 ------------------------------------------------
    def table_transition_rows(self, d):
        n = self.lib.orc_table_transition_rows(self.h, d, None, 0)
        rows = np.zeros((n, 4), np.uint64)
        self.lib.orc_table_transition_rows(self.h, d, _p(rows, _u64p), n)
        return rows

    def table_endpoint_rows(self, d):
        n = self.lib.orc_table_endpoint_rows(self.h, d, None, 0)
        rows = np.zeros((n, 3), np.uint64)
        self.lib.orc_table_endpoint_rows(self.h, d, _p(rows, _u64p), n)
        return rows


def reference_cases():
    return json.load(open(os.path.join(GOLDEN, "reference_tests.json")))


def decode_status(s):
    s = int(s)
    code = s & 0xff
    if code == 0:
        return {"code": 0, "accept": (s >> 8) & 0xffffffff}
    if code == 1:
        return {"code": 1, "def": (s >> 8) & 0xff, "char": (s >> 16) & 0xff, "state": (s >> 24) & 0xffff, "pos": s >> 40}
    if code == 2:
        return {"code": 2, "pos": s >> 40}
    return {"code": code}


def oracle_fr_from_u64(lib, v):
    """F::from(v), F = bn256::Fr, by the oracle's restatement of halo2curves' Montgomery multiplication: 4 u64 limbs."""
    out = (C.c_uint64 * 4)()
    lib.orc_fr_from_u64(int(v), out)
    return [int(x) for x in out]
