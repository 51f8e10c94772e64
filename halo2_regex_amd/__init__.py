"""halo2_regex_amd — MI355X-native batched DFA witness generator for the halo2-regex chip.

Python here is a thin ctypes binding over the C ABI in include/hrx.h (libhrx.so, built from
halo2_regex_amd/csrc by hipcc for gfx950).  The class and method names mirror the reference's
Rust surface for the path (src/defs.rs, src/table.rs, src/lib.rs:96-113,311-318,766-888) so that the
parity tests read like the reference's own tests.  There is no CPU implementation of the compute
path: without the HIP library the import fails, and without a gfx950 device RegexVerifyConfig raises.
"""
import ctypes as C
import os

# torch must be imported before libhrx.so: both need libamdhip64.so.7 and the process must end up with
# ONE HIP runtime (torch's bundled copy wins by SONAME when it is loaded first), otherwise device
# pointers of torch tensors would belong to a different runtime than the one launching the kernels.
import torch  # noqa: F401  (plumbing: device memory, streams, torch.distributed)
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HRX_LIB_PATH") or os.path.join(_HERE, "csrc", "libhrx.so")   # (HRX_LIB_PATH: tools/ load the ablation build)

HRX_OK = 0
HRX_ERR_PARSE, HRX_ERR_BOUNDS, HRX_ERR_ARG, HRX_ERR_HIP, HRX_ERR_STATE = 1, 2, 3, 4, 5
HRX_ERR_INVALID_TRANSITION, HRX_ERR_OUT_OF_CONTRACT, HRX_ERR_IO = 6, 7, 8
HRX_DEVICE_SAME = -2                 # hrx_ctx_clone: the source context's device
HRX_DEVICE_NONE = -1                 # hrx_ctx_create: host-only context (the native small-batch host walk)
HRX_DEFAULT_HOST_THRESHOLD = 32768   # rows (B x M) below which host-buffer batches are walked on the host


_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_u16p = C.POINTER(C.c_uint16)
_u8p = C.POINTER(C.c_uint8)


class _RegexPartC(C.Structure):      # hrx_regex_part of include/hrx.h
    _fields_ = [("regex_def", C.c_char_p), ("regex_len", C.c_size_t), ("is_public", C.c_int), ("max_size", C.c_size_t)]


class _HostRouteReportC(C.Structure):    # hrx_host_route_report of include/hrx.h
    _fields_ = [("route", C.c_int), ("device_strings", C.c_size_t), ("host_strings", C.c_size_t), ("device_ms", C.c_double), ("host_ms", C.c_double), ("call_ms", C.c_double),
                ("device_alone_ns_per_row", C.c_double), ("host_alone_ns_per_row", C.c_double), ("split_ns_per_row", C.c_double),
                ("device_ns_per_row", C.c_double), ("host_ns_per_row", C.c_double), ("host_threads", C.c_int), ("device_pipelined", C.c_int)]


class _PlaceReportC(C.Structure):    # hrx_place_report of include/hrx.h
    _fields_ = [("searched", C.c_int), ("steps", C.c_int), ("accepted", C.c_int), ("chosen_step", C.c_int),
                ("ref_us", C.c_double), ("first_us", C.c_double), ("best_us", C.c_double),
                ("ref_gbs", C.c_double), ("first_gbs", C.c_double), ("best_gbs", C.c_double), ("probe_bytes", C.c_size_t),
                ("peak_candidate_bytes", C.c_size_t), ("search_ms", C.c_double), ("capped", C.c_int)]


class HrxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "halo2_regex_amd: %s is missing — build it with `make -C halo2_regex_amd/csrc` "
            "(or __graft_entry__.build()); there is no fallback path" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, sz, u64, i = C.c_void_p, C.c_size_t, C.c_uint64, C.c_int
    sig = {
        "hrx_defs_create": (i, [C.POINTER(vp)]),
        "hrx_defs_destroy": (None, [vp]),
        "hrx_defs_push_allstr_text": (i, [vp, C.c_char_p, sz]),
        "hrx_defs_push_allstr_file": (i, [vp, C.c_char_p]),
        "hrx_defs_push_substr_text": (i, [vp, C.c_char_p, sz]),
        "hrx_defs_push_substr_file": (i, [vp, C.c_char_p]),
        "hrx_defs_push_allstr": (i, [vp, u64, u64, u64, sz, _u64p, _u64p, _u8p, _u64p]),
        "hrx_defs_push_substr": (i, [vp, sz, _u64p, _u64p, sz, _u64p, sz, _u64p]),
        "hrx_defs_finalize": (i, [vp]),
        "hrx_defs_num_defs": (sz, [vp]),
        "hrx_defs_num_substrs": (sz, [vp, sz]),
        "hrx_defs_first_state": (u64, [vp, sz]),
        "hrx_defs_accepted_state": (u64, [vp, sz]),
        "hrx_defs_largest_state": (u64, [vp, sz]),
        "hrx_defs_num_transitions": (sz, [vp, sz]),
        "hrx_defs_substr_id_offset": (u64, [vp, sz]),
        "hrx_defs_table_bytes": (sz, [vp]),
        "hrx_table_transition_rows": (sz, [vp, sz, _u64p, sz]),
        "hrx_table_endpoint_rows": (sz, [vp, sz, _u64p, sz]),
        "hrx_device_count": (i, [C.POINTER(i)]),
        "hrx_ctx_create": (i, [vp, i, C.POINTER(vp)]),
        "hrx_ctx_destroy": (None, [vp]),
        "hrx_ctx_device": (i, [vp]),
        "hrx_ctx_clone": (i, [vp, i, C.POINTER(vp)]),
        "hrx_ctx_set_host_threshold": (i, [vp, sz]),
        "hrx_ctx_host_threshold": (sz, [vp]),
        "hrx_ctx_set_placement": (i, [vp, i, sz, C.c_double]),
        "hrx_last_error": (C.c_char_p, []),
        "hrx_witness_batch_device": (i, [vp, vp, sz, vp, sz, sz, vp, vp, vp, vp]),
        "hrx_witness_batch_device_pitched": (i, [vp, vp, sz, vp, sz, sz, vp, sz, vp, sz, vp, vp]),
        "hrx_recommended_pitches": (None, [sz, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]),
        "hrx_witness_batch_device_layout": (i, [vp, i, vp, sz, vp, sz, sz, vp, vp, vp, vp]),
        "hrx_position_major_sizes": (None, [sz, sz, sz, C.POINTER(sz), C.POINTER(sz)]),
        "hrx_witness_batch_device_planes": (i, [vp, i, vp, sz, vp, sz, sz, C.POINTER(vp), sz, vp, vp, vp]),
        "hrx_position_major_plane_sizes": (None, [sz, sz, C.POINTER(sz), C.POINTER(sz)]),
        "hrx_position_major_stripe_sizes": (None, [sz, sz, sz, C.POINTER(sz), C.POINTER(sz)]),
        "hrx_alloc_output_planes": (i, [vp, sz, sz, sz, C.POINTER(vp), C.POINTER(vp)]),
        "hrx_alloc_output_planes_for_batch": (i, [vp, i, vp, sz, vp, sz, sz, sz, C.POINTER(vp), C.POINTER(vp)]),
        "hrx_rows_of_string_planes": (i, [C.POINTER(vp), sz, vp, sz, sz, sz, sz, vp, vp]),
        "hrx_probe_write_pair": (i, [vp, vp, vp, sz, C.POINTER(C.c_double)]),
        "hrx_traffic_pass_device_planes": (i, [vp, vp, sz, sz, sz, C.POINTER(vp), sz, vp, vp]),
        "hrx_witness_of_string": (i, [vp, sz, sz, sz, vp, vp, vp, vp]),
        "hrx_witness_num_columns": (sz, [sz]),
        "hrx_witness_columns_host": (i, [i, vp, sz, vp, vp, sz, vp, sz, sz, sz, sz, sz, sz, vp]),
        "hrx_ctx_host_route_report": (i, [vp, C.POINTER(_HostRouteReportC), sz]),
        "hrx_alloc_last_report_sized": (i, [vp, vp, sz]),
        "hrx_ctx_set_option": (i, [vp, i, C.c_long]),
        "hrx_ctx_get_option": (C.c_long, [vp, i]),
        "hrx_rows_of_string_position_major": (i, [vp, vp, sz, sz, sz, sz, vp, vp]),
        "hrx_describe_launch": (i, [vp, i, sz, sz, i, C.c_char_p, sz]),
        "hrx_ctx_describe_launch": (i, [vp, i, sz, sz, C.c_char_p, sz]),
        "hrx_alloc_outputs_position_major": (i, [vp, sz, sz, C.POINTER(vp), C.POINTER(vp)]),
        "hrx_alloc_output_pair": (i, [vp, sz, sz, C.POINTER(vp), C.POINTER(vp)]),
        "hrx_device_free": (i, [vp]),
        "hrx_alloc_last_report": (i, [vp, C.POINTER(_PlaceReportC)]),
        "hrx_traffic_pass_device": (i, [vp, vp, sz, sz, sz, vp, vp, vp]),
        "hrx_traffic_pass_device_layout": (i, [vp, i, vp, sz, sz, sz, vp, sz, vp, sz, vp]),
        "hrx_chars_to_position_major_device": (i, [vp, vp, sz, sz, vp, vp]),
        "hrx_multi_create": (i, [vp, C.POINTER(i), i, C.POINTER(vp)]),
        "hrx_multi_destroy": (None, [vp]),
        "hrx_multi_num_shards": (i, [vp]),
        "hrx_multi_shard_device": (i, [vp, i]),
        "hrx_multi_shard_stream": (vp, [vp, i]),
        "hrx_multi_witness_batch_device": (i, [vp, i, C.POINTER(vp), sz, C.POINTER(vp), C.POINTER(sz), sz, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
        "hrx_multi_synchronize": (i, [vp]),
        "hrx_multi_witness_batch_host": (i, [vp, _u8p, sz, _u32p, sz, sz, _u32p, _u16p, _u64p]),
        "hrx_fr_num_columns": (sz, [sz]),
        "hrx_fr_columns_device": (i, [vp, i, vp, sz, vp, vp, sz, vp, sz, sz, sz, sz, sz, vp, i, vp]),
        "hrx_fr_columns_device_planes": (i, [vp, i, vp, sz, vp, C.POINTER(vp), sz, vp, sz, sz, sz, sz, vp, i, vp]),
        "hrx_fr_from_u64": (None, [C.c_uint64, i, _u64p]),
        "hrx_witness_batch_host": (i, [vp, _u8p, sz, _u32p, sz, sz, _u32p, _u16p, _u64p]),
        "hrx_shard_range": (None, [sz, i, i, C.POINTER(sz), C.POINTER(sz)]),
        "hrx_derive_states": (i, [vp, _u8p, sz, _u64p]),
        "hrx_derive_substr_ids": (i, [vp, _u64p, sz, _u64p]),
        "hrx_derive_is_start_end": (i, [vp, _u64p, _u64p, sz, _u8p, _u8p]),
        "hrx_match_substrs": (i, [vp, _u8p, sz, sz] + [_u64p] * 9),
        "hrx_regex_to_allstr_text": (i, [C.c_char_p, sz, C.c_char_p, sz, C.POINTER(sz)]),
        "hrx_regex_to_dfa_json": (i, [C.c_char_p, sz, C.c_char_p, sz, C.POINTER(sz)]),
        "hrx_gen_regex_files": (i, [C.POINTER(_RegexPartC), sz, sz, C.POINTER(vp)]),
        "hrx_regex_files_num_substrs": (sz, [vp]),
        "hrx_regex_files_allstr": (vp, [vp, C.POINTER(sz)]),
        "hrx_regex_files_substr": (vp, [vp, sz, C.POINTER(sz)]),
        "hrx_regex_files_destroy": (None, [vp]),
        "hrx_format_regex_str": (i, [C.c_char_p, sz, C.c_char_p, sz, C.POINTER(sz)]),
        "hrx_regex_find": (i, [C.c_char_p, sz, C.c_char_p, sz, C.POINTER(i), C.POINTER(sz), C.POINTER(sz)]),
    }
    _load.symbols = list(sig)
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = args
    if hasattr(lib, "hrx_chunked_alloc"):      # tools builds only (libhrx_ablation.so: csrc/hrx_alloc.cpp, the placement probes)
        lib.hrx_chunked_alloc.restype, lib.hrx_chunked_alloc.argtypes = i, [i, sz, C.POINTER(vp)]
        lib.hrx_chunked_free.restype, lib.hrx_chunked_free.argtypes = i, [vp]
    return lib


lib = _load()
#: every symbol include/hrx.h declares = every symbol bound above (tests check that the header, this list and the library's exports agree)
ABI_SYMBOLS = list(_load.symbols)


class DeviceBuffer:
    """TOOLS ONLY (needs libhrx_ablation.so): `nbytes` of device memory from csrc/hrx_alloc.cpp — a virtual range over 2-MiB
    physical chunks, for the placement probes of DESIGN.md §6 — exposed through __cuda_array_interface__; freed with the object."""

    def __init__(self, nbytes, device=0):
        if not hasattr(lib, "hrx_chunked_alloc"):
            raise HrxError(HRX_ERR_STATE, "DeviceBuffer needs the tools build: HRX_LIB_PATH=.../libhrx_ablation.so")
        p = C.c_void_p()
        _check(lib.hrx_chunked_alloc(int(device), int(nbytes), C.byref(p)))
        self.ptr, self.nbytes, self.device = p.value, int(nbytes), int(device)
        self.__cuda_array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2}

    def tensor(self, dtype=None):
        """a torch view of the whole buffer (keeps this object alive)"""
        t = torch.as_tensor(self, device=torch.device("cuda", self.device))
        return t if dtype is None else t.view(dtype)

    def __del__(self):
        if getattr(self, "ptr", None) and lib is not None:      # (lib is None while the interpreter shuts down)
            lib.hrx_chunked_free(self.ptr)
            self.ptr = None


class _LibraryOwned:
    """device memory handed out by the library (hrx_alloc_outputs_position_major), as a __cuda_array_interface__ object: the
    torch tensors made from it keep it alive, hrx_device_free runs when the last of them goes"""

    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, int(nbytes)
        self.__cuda_array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2}

    def __del__(self):
        if getattr(self, "ptr", None) and lib is not None:
            lib.hrx_device_free(self.ptr)
            self.ptr = None


def device_empty(numel, dtype, device, chunked=False):
    """torch.empty((numel,), dtype) on `device`; chunked=True (tools only): backed by a DeviceBuffer"""
    dev = torch.device(device)
    if not chunked:
        return torch.empty((int(numel),), dtype=dtype, device=dev)
    nbytes = max(int(numel) * torch.empty((), dtype=dtype).element_size(), 1)
    return DeviceBuffer(nbytes, dev.index if dev.index is not None else torch.cuda.current_device()).tensor(dtype)[:numel]


def _check(rc):
    if rc != HRX_OK:
        raise HrxError(rc, lib.hrx_last_error().decode("utf-8", "replace"))


def _np(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _ptr(a, t):
    return a.ctypes.data_as(t)


# ---------------------------------------------------------------------------------------------
# definition generation — src/vrm/mod.rs, src/vrm/js_caller.rs, src/vrm/regex.js (host only)
# ---------------------------------------------------------------------------------------------
def _two_call(fn, regex):
    data = regex.encode("utf-8") if isinstance(regex, str) else bytes(regex)
    need = C.c_size_t(0)
    _check(fn(data, len(data), None, 0, C.byref(need)))
    buf = C.create_string_buffer(max(need.value, 1))
    _check(fn(data, len(data), buf, need.value, C.byref(need)))
    return buf.raw[:need.value]


def get_dfa_json_value(regex):
    """get_dfa_json_value (src/vrm/js_caller.rs:43-48): the minimal DFA of `regex` as the parsed JSON value
    regexToDfa returns — computed natively, no JS engine."""
    import json
    return json.loads(_two_call(lib.hrx_regex_to_dfa_json, regex).decode("utf-8"))


def regex_to_dfa_json_text(regex):
    return _two_call(lib.hrx_regex_to_dfa_json, regex).decode("utf-8")


def regex_to_allstr_text(regex):
    """regexToDfa + dfa_to_regex_def_text (src/vrm/js_caller.rs:127-157): the AllstrRegexDef text of `regex`."""
    return _two_call(lib.hrx_regex_to_allstr_text, regex).decode("ascii")


def format_regex_str(regex):
    """format_regex_str (src/vrm/js_caller.rs:36-41): formatRegexPrintable of src/vrm/regex.js:24-39."""
    return _two_call(lib.hrx_format_regex_str, regex).decode("utf-8")


def regex_find(pattern, text):
    """The leftmost-first search gen_regex_files runs for every part on every DFA path (mod.rs:553-583);
    returns (start, end) or None."""
    p = pattern.encode("utf-8") if isinstance(pattern, str) else bytes(pattern)
    t = text.encode("utf-8") if isinstance(text, str) else bytes(text)
    found, s0, e0 = C.c_int(0), C.c_size_t(0), C.c_size_t(0)
    _check(lib.hrx_regex_find(p, len(p), t, len(t), C.byref(found), C.byref(s0), C.byref(e0)))
    return (s0.value, e0.value) if found.value else None


class RegexPartConfig:
    """RegexPartConfig (src/vrm/mod.rs:39-49)."""

    def __init__(self, is_public, regex_def, max_size, solidity=None):
        self.is_public, self.regex_def, self.max_size, self.solidity = bool(is_public), regex_def, int(max_size), solidity


class DecomposedRegexConfig:
    """DecomposedRegexConfig (src/vrm/mod.rs:31-37): gen_regex_files (mod.rs:62-307) computed natively — the
    AllstrRegexDef text of the concatenated parts and one SubstrRegexDef text per public part."""

    def __init__(self, max_byte_size, parts):
        self.max_byte_size, self.parts = int(max_byte_size), list(parts)

    @classmethod
    def from_json(cls, text):
        import json
        v = json.loads(text)
        return cls(v["max_byte_size"], [RegexPartConfig(p["is_public"], p["regex_def"], p["max_size"], p.get("solidity"))
                                       for p in v["parts"]])

    def all_regex(self):
        return "".join(p.regex_def for p in self.parts)   # mod.rs:87-91

    def gen_allstr_text(self):
        return regex_to_allstr_text(self.all_regex())

    def gen_allstr_file(self, allstr_file_path):
        with open(allstr_file_path, "w") as f:
            f.write(self.gen_allstr_text())

    def gen_regex_texts(self):
        """-> (allstr_text, [substr_text per public part])"""
        keep = [p.regex_def.encode("utf-8") for p in self.parts]
        arr = (_RegexPartC * max(len(keep), 1))()
        for k, (p, b) in enumerate(zip(self.parts, keep)):
            arr[k] = _RegexPartC(b, len(b), int(p.is_public), p.max_size)
        h = C.c_void_p()
        _check(lib.hrx_gen_regex_files(arr, len(keep), self.max_byte_size, C.byref(h)))
        try:
            n = C.c_size_t(0)
            ptr = lib.hrx_regex_files_allstr(h, C.byref(n))
            allstr = C.string_at(ptr, n.value).decode("ascii")
            subs = []
            for k in range(lib.hrx_regex_files_num_substrs(h)):
                ptr = lib.hrx_regex_files_substr(h, k, C.byref(n))
                subs.append(C.string_at(ptr, n.value).decode("ascii"))
        finally:
            lib.hrx_regex_files_destroy(h)
        return allstr, subs

    def gen_regex_files(self, allstr_file_path, substr_file_pathes):
        """gen_regex_files (mod.rs:62-307): writes the AllstrRegexDef file and one SubstrRegexDef file per public part."""
        allstr, subs = self.gen_regex_texts()
        if len(substr_file_pathes) < len(subs):
            raise ValueError("%d public parts but %d substr paths" % (len(subs), len(substr_file_pathes)))
        with open(allstr_file_path, "w") as f:
            f.write(allstr)
        for path, text in zip(substr_file_pathes, subs):
            with open(path, "w") as f:
                f.write(text)


# ---------------------------------------------------------------------------------------------
# data model — src/defs.rs
# ---------------------------------------------------------------------------------------------
class AllstrRegexDef:
    """AllstrRegexDef (src/defs.rs:26-36): holds the definition text; parsing happens in C."""

    def __init__(self, text):
        self.text = text.encode() if isinstance(text, str) else bytes(text)

    @classmethod
    def read_from_text(cls, file_path):  # defs.rs:54-58
        with open(file_path, "rb") as f:
            return cls(f.read())


class SubstrRegexDef:
    """SubstrRegexDef (src/defs.rs:115-132)."""

    def __init__(self, text):
        self.text = text.encode() if isinstance(text, str) else bytes(text)

    @classmethod
    def read_from_text(cls, file_path):  # defs.rs:184-188
        with open(file_path, "rb") as f:
            return cls(f.read())

    @classmethod
    def new(cls, max_length, min_position, max_position, valid_state_transitions, start_states, end_states):
        """SubstrRegexDef::new (defs.rs:147-163), rendered to the text format of defs.rs:165-177."""
        lines = [str(max_length), str(min_position), str(max_position),
                 " ".join(str(s) for s in start_states), " ".join(str(s) for s in end_states)]
        lines += ["%d %d" % (a, b) for a, b in sorted(valid_state_transitions)]
        return cls("\n".join(lines) + "\n")


class RegexDefs:
    """RegexDefs { allstr, substrs } (src/defs.rs:17-22)."""

    def __init__(self, allstr, substrs):
        self.allstr = allstr
        self.substrs = list(substrs)


class _DefsHandle:
    def __init__(self, regex_defs):
        self.h = C.c_void_p()
        _check(lib.hrx_defs_create(C.byref(self.h)))
        try:
            for rd in regex_defs:
                _check(lib.hrx_defs_push_allstr_text(self.h, rd.allstr.text, len(rd.allstr.text)))
                for sd in rd.substrs:
                    _check(lib.hrx_defs_push_substr_text(self.h, sd.text, len(sd.text)))
            _check(lib.hrx_defs_finalize(self.h))
        except Exception:
            lib.hrx_defs_destroy(self.h)
            self.h = None
            raise

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:   # (module globals are gone at interpreter shutdown)
            lib.hrx_defs_destroy(self.h)
            self.h = None


class RegexTableConfig:
    """The integer rows RegexTableConfig::load assigns (src/table.rs:61-198)."""

    def __init__(self, defs_handle, def_idx):
        self._d = defs_handle
        self.def_idx = def_idx

    def transition_rows(self):
        n = lib.hrx_table_transition_rows(self._d.h, self.def_idx, None, 0)
        rows = np.zeros((n, 4), np.uint64)
        lib.hrx_table_transition_rows(self._d.h, self.def_idx, _ptr(rows, _u64p), n)
        return rows

    def endpoint_rows(self):
        n = lib.hrx_table_endpoint_rows(self._d.h, self.def_idx, None, 0)
        rows = np.zeros((n, 3), np.uint64)
        lib.hrx_table_endpoint_rows(self._d.h, self.def_idx, _ptr(rows, _u64p), n)
        return rows


class AssignedRegexResult:
    """Integer content of AssignedRegexResult (src/lib.rs:79-93) plus the per-def advice columns."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def device_count():
    n = C.c_int(0)
    _check(lib.hrx_device_count(C.byref(n)))
    return n.value


def recommended_pitches(M):
    """(records pitch, masked pitch) in rows and an input stride in bytes that keep the write/read fronts off a
    power-of-two stride (hrx_recommended_pitches)."""
    a, b, c = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    lib.hrx_recommended_pitches(M, C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


LAYOUT_STRING_MAJOR, LAYOUT_POSITION_MAJOR, LAYOUT_INPUT_POSITION_MAJOR = 0, 1, 2
LAYOUT_RECORD_PLANES = 4      # describe_launch only: the launch witness_batch_planes makes


PLACE_OFF, PLACE_WALK = 0, 1     # hrx_ctx_set_placement modes
OPT_PMD_COMBINER_WAVE = 1        # hrx_ctx_set_option: 0 default, 1 on, 2 off
OPT_HOST_ROUTE, OPT_HOST_THREADS, OPT_HOST_PIPELINE, OPT_PLACE_DRY_LAUNCH = 2, 3, 4, 5
HOST_ROUTE_AUTO, HOST_ROUTE_DEVICE, HOST_ROUTE_HOST = 0, 1, 2
PLACE_CAPPED_STEPS, PLACE_CAPPED_BYTES, PLACE_CAPPED_TIME, PLACE_CAPPED_ALLOC = 1, 2, 4, 8      # hrx_place_report.capped
PLACED_FROM = 128 << 20    # alloc_outputs*: records of this many bytes or more come from the library's placement-aware allocator (kPlaceFromBytes)
PM_BLOCK = 65536          # kPmBlock of csrc/hrx_lane.h: position-major buffers are blocked by this many strings


def _cat(parts):
    return parts[0] if len(parts) == 1 else (torch.cat(parts) if hasattr(parts[0], "permute") else np.concatenate(parts))


def chars_to_position_major(chars):
    """(B, stride) bytes, stride % 16 == 0  ->  flat HRX_LAYOUT_INPUT_POSITION_MAJOR buffer: per block of PM_BLOCK strings
    [stride/16][nb][16], blocks back to back (one block = the plain layout for B <= PM_BLOCK); torch or numpy."""
    B, stride = chars.shape
    if hasattr(chars, "is_cuda") and chars.is_cuda:       # written block by block into the one output buffer (no second copy of the batch)
        out = torch.empty((B * stride,), dtype=chars.dtype, device=chars.device)
        for k0 in range(0, B, PM_BLOCK):
            c = chars[k0:k0 + PM_BLOCK]
            nb = c.shape[0]
            out[k0 * stride:(k0 + nb) * stride].view(stride // 16, nb, 16).copy_(c.reshape(nb, stride // 16, 16).permute(1, 0, 2))
        return out
    parts = []
    for k0 in range(0, B, PM_BLOCK):
        c = chars[k0:k0 + PM_BLOCK]
        c = c.reshape(c.shape[0], stride // 16, 16)
        c = c.permute(1, 0, 2).contiguous() if hasattr(c, "permute") else np.ascontiguousarray(c.transpose(1, 0, 2))
        parts.append(c.reshape(-1))
    return _cat(parts)


def position_major_to_string_major(records_pm, masked_pm, B, M, D):
    """Inverse of HRX_LAYOUT_POSITION_MAJOR (include/hrx.h): per block of PM_BLOCK strings records [ceil(M/4)][D][nb][4] ->
    (nb, M, D), masked [ceil(M/8)][nb][8] -> (nb, M); returns (B, M, D) and (B, M).  Works on torch tensors and numpy arrays
    alike; no values change (a view for a single block)."""
    q4, q8 = (M + 3) // 4, (M + 7) // 8
    recs, msks = [], []
    for k0 in range(0, B, PM_BLOCK):
        nb = min(PM_BLOCK, B - k0)
        r = records_pm[k0 * q4 * 4 * D:(k0 + nb) * q4 * 4 * D].reshape(q4, D, nb, 4)
        m = masked_pm[k0 * q8 * 8:(k0 + nb) * q8 * 8].reshape(q8, nb, 8)
        r = r.permute(2, 0, 3, 1) if hasattr(r, "permute") else r.transpose(2, 0, 3, 1)      # (nb, M/4, 4, D)
        m = m.permute(1, 0, 2) if hasattr(m, "permute") else m.transpose(1, 0, 2)
        recs.append(r.reshape(nb, -1, D)[:, :M])
        msks.append(m.reshape(nb, -1)[:, :M])
    return _cat(recs), _cat(msks)


def rows_of_string_position_major(records_pm, masked_pm, B, M, D, b, out=None):
    """hrx_rows_of_string_position_major: one circuit's rows — (M, D) uint32 records and (M,) uint16 masked rows of string b — gathered on the host out of
    position-major HOST arrays (numpy; e.g. the device buffers copied out as they are)."""
    rec = np.empty((M, D), np.uint32) if out is None else out[0]
    msk = np.empty(M, np.uint16) if out is None else out[1]
    rp = np.ascontiguousarray(records_pm).view(np.uint32)
    mp = np.ascontiguousarray(masked_pm).view(np.uint16)
    _check(lib.hrx_rows_of_string_position_major(rp.ctypes.data, mp.ctypes.data, B, M, D, b, rec.ctypes.data, msk.ctypes.data))
    return rec, msk


def planes_to_string_major(planes, masked_pm, B, M, D=None):
    """Inverse of the record-planes layout (include/hrx.h hrx_witness_batch_device_planes): the D planes — each per block of PM_BLOCK strings [ceil(M/4)][nb][4] — or, D = 1 with two
    buffers, the two ROW STRIPES of the one def (quad q in buffer q % 2 at slot q / 2), and the position-major masked rows -> (B, M, D) and (B, M); torch tensors or numpy arrays."""
    D = len(planes) if D is None else D
    R = len(planes) // D
    q4, q8 = (M + 3) // 4, (M + 7) // 8
    slots = (q4 + R - 1) // R
    is_t = hasattr(planes[0], "permute")
    recs, msks = [], []
    for k0 in range(0, B, PM_BLOCK):
        nb = min(PM_BLOCK, B - k0)
        per_def = []
        for d in range(D):
            quads = []      # (slots, nb, 4) per stripe -> interleave the stripes' slots back into quad order
            for r in range(R):
                x = planes[r * D + d][k0 * slots * 4:(k0 + nb) * slots * 4].reshape(slots, nb, 4)
                quads.append(x)
            if R == 1:
                q = quads[0]
            else:
                q = (torch.stack(quads, dim=1) if is_t else np.stack(quads, axis=1)).reshape(slots * R, nb, 4)[:q4]
            q = q.permute(1, 0, 2) if is_t else q.transpose(1, 0, 2)            # (nb, q4, 4)
            per_def.append(q.reshape(nb, -1)[:, :M])
        recs.append(torch.stack(per_def, dim=2) if is_t else np.stack(per_def, axis=2))
        m = masked_pm[k0 * q8 * 8:(k0 + nb) * q8 * 8].reshape(q8, nb, 8)
        m = m.permute(1, 0, 2) if is_t else m.transpose(1, 0, 2)
        msks.append(m.reshape(nb, -1)[:, :M])
    return _cat(recs), _cat(msks)


def rows_of_string_planes(planes, masked_pm, B, M, b, out=None, D=None):
    """hrx_rows_of_string_planes: one circuit's rows out of record planes (or the two row stripes of one def: D=1) in HOST memory (numpy arrays, each buffer copied from the device as it is)."""
    D = len(planes) if D is None else D
    rec = np.empty((M, D), np.uint32) if out is None else out[0]
    msk = np.empty(M, np.uint16) if out is None else out[1]
    ps = [np.ascontiguousarray(p).view(np.uint32) for p in planes]
    arr = (C.c_void_p * len(ps))(*[p.ctypes.data for p in ps])
    mp = np.ascontiguousarray(masked_pm).view(np.uint16)
    _check(lib.hrx_rows_of_string_planes(arr, len(ps), mp.ctypes.data, B, M, D, b, rec.ctypes.data, msk.ctypes.data))
    return rec, msk


def witness_of_string(records, n):
    """hrx_witness_of_string: (states (D, n+1) uint64, substr_ids (D, n) uint64, is_starts (D, n+1) bool, is_ends (D, n+1) bool) — what lib.rs:316-318 derive —
    out of one string's compact records, a (M, D) uint32 array."""
    rec = np.ascontiguousarray(records).view(np.uint32)
    M, D = rec.shape
    states, sids = np.zeros((D, n + 1), np.uint64), np.zeros((D, n), np.uint64)      # (usize == u64 on every target of this library)
    st, en = np.zeros((D, n + 1), np.uint8), np.zeros((D, n + 1), np.uint8)
    _check(lib.hrx_witness_of_string(rec.ctypes.data, D, n, M, states.ctypes.data, sids.ctypes.data, st.ctypes.data, en.ctypes.data))
    return states, sids, st.astype(bool), en.astype(bool)


def witness_columns_host(chars, lens, records, masked, M, D, b_begin=0, b_count=None, position_major=False, chars_pm_stride=None, B=None):
    """hrx_witness_columns_host: (4 + 4 D, b_count, M) uint64 — the integer content of every advice column of circuits [b_begin, b_begin + b_count) out of a finished batch
    in host memory (numpy): string-major records (B, M, D) / masked (B, M) / chars (B, stride), or the flat position-major buffers with position_major=True."""
    lens = np.ascontiguousarray(lens, np.uint32)
    B = len(lens) if B is None else B
    b_count = B - b_begin if b_count is None else b_count
    cols = np.empty((4 + 4 * D, b_count, M), np.uint64)
    rec, msk, ch = np.ascontiguousarray(records).view(np.uint32), np.ascontiguousarray(masked).view(np.uint16), np.ascontiguousarray(chars)
    if position_major:
        layout = LAYOUT_POSITION_MAJOR | (LAYOUT_INPUT_POSITION_MAJOR if chars_pm_stride else 0)
        stride = int(chars_pm_stride) if chars_pm_stride else ch.shape[1]
        rp = mp = 0
    else:
        layout, stride, rp, mp = LAYOUT_STRING_MAJOR, ch.shape[1], rec.strides[0] // (4 * D), msk.strides[0] // 2
    _check(lib.hrx_witness_columns_host(layout, ch.ctypes.data, stride, lens.ctypes.data, rec.ctypes.data, rp, msk.ctypes.data, mp, B, M, D, b_begin, b_count, cols.ctypes.data))
    return cols


def shard_range(B, world, rank):
    b, c = C.c_size_t(0), C.c_size_t(0)
    lib.hrx_shard_range(B, world, rank, C.byref(b), C.byref(c))
    return b.value, c.value


class RegexVerifyConfig:
    """Witness-side mirror of RegexVerifyConfig (src/lib.rs:96-113).

    configure() takes what the reference's configure (lib.rs:126-131) takes minus the halo2 objects
    (ConstraintSystem, FlexGateConfig): max_chars_size and the Vec<RegexDefs>.  `device=None` builds the
    host-side tables only (table rows, constants); any compute call then raises.  `device=HRX_DEVICE_NONE`
    makes a host-only context: match_substrs / derive_* / witness_batch_host run the library's native
    small-batch host walk (what a drop-in for the one-string-per-call lib.rs:316-318 needs), device batches raise.
    """

    def __init__(self, max_chars_size, regex_defs, device=0):
        self.max_chars_size = int(max_chars_size)
        self.regex_defs = list(regex_defs)      # pub field, lib.rs:112
        self._defs = _DefsHandle(self.regex_defs)
        self.num_defs = lib.hrx_defs_num_defs(self._defs.h)
        self.table_array = [RegexTableConfig(self._defs, d) for d in range(self.num_defs)]
        self._ctx = None
        self.device = device
        if device is not None:   # a GPU index, or HRX_DEVICE_NONE for the host-only context
            ctx = C.c_void_p()
            _check(lib.hrx_ctx_create(self._defs.h, int(device), C.byref(ctx)))
            self._ctx = ctx

    @classmethod
    def configure(cls, max_chars_size, regex_defs, device=0):
        return cls(max_chars_size, regex_defs, device)

    def __del__(self):
        if getattr(self, "_ctx", None) and lib is not None:
            lib.hrx_ctx_destroy(self._ctx)
            self._ctx = None

    def clone(self, device=HRX_DEVICE_SAME):
        """hrx_ctx_clone: RegexVerifyConfig derives Clone (lib.rs:96) — the same config with a context (stream, scratch, lock) of its own, e.g. one per prover thread."""
        import copy
        ctx = C.c_void_p()
        _check(lib.hrx_ctx_clone(self._need_ctx(), int(device), C.byref(ctx)))   # first: a failed clone must leave nothing behind that owns the source's context
        c = copy.copy(self)
        c._ctx = ctx
        c.device = self.device if device == HRX_DEVICE_SAME else device
        return c

    def _need_ctx(self):
        if not self._ctx:
            raise HrxError(HRX_ERR_HIP, "no context: configure(..., device=k) for a gfx950 GPU or device=HRX_DEVICE_NONE for the host walk")
        return self._ctx

    def _need_device(self, *tensors):
        """The batch kernels run on this config's device: every tensor must live there."""
        ctx = self._need_ctx()
        for t in tensors:
            if t is not None and (not t.is_cuda or t.device.index != self.device):
                raise HrxError(HRX_ERR_ARG, "tensor on %s but the config's context is on cuda:%s" % (t.device, self.device))
        return ctx

    def set_host_threshold(self, rows):
        """hrx_ctx_set_host_threshold: host-buffer batches of fewer than `rows` witness rows take the host walk."""
        _check(lib.hrx_ctx_set_host_threshold(self._need_ctx(), int(rows)))

    def host_threshold(self):
        return lib.hrx_ctx_host_threshold(self._need_ctx())

    def set_placement(self, walk=True, max_bytes=0, max_ms=0.0):
        """hrx_ctx_set_placement: placement-aware output allocation of this config's context off / on, and the memory / time budget of one walk (0 = default)."""
        _check(lib.hrx_ctx_set_placement(self._need_ctx(), PLACE_WALK if walk else PLACE_OFF, int(max_bytes), float(max_ms)))

    # -- lookup tables: RegexVerifyConfig::load (lib.rs:779-785) ---------------------------------
    def load(self):
        """Returns per def (transition_rows, endpoint_rows) in table.rs assignment order."""
        return [(t.transition_rows(), t.endpoint_rows()) for t in self.table_array]

    def substr_id_offset(self, d):
        return lib.hrx_defs_substr_id_offset(self._defs.h, d)

    def accepted_state(self, d):
        return lib.hrx_defs_accepted_state(self._defs.h, d)

    def first_state(self, d):
        return lib.hrx_defs_first_state(self._defs.h, d)

    def table_bytes(self):
        return lib.hrx_defs_table_bytes(self._defs.h)

    def describe_launch(self, B, layout=0, num_cus=256):
        """Kernel name and launch geometry the planner picks for B strings in `layout` (include/hrx.h: 0 string-major, 1 position-major
        outputs, 3 position-major input and outputs); host-only; MI355X has 256 CUs."""
        buf = C.create_string_buffer(4096)
        if self._ctx and num_cus == 256:    # the context's own view: its device's CUs, the flags it was created with, its set_option choices
            _check(lib.hrx_ctx_describe_launch(self._ctx, layout, B, self.max_chars_size, buf, 4096))
        else:
            _check(lib.hrx_describe_launch(self._defs.h, layout, B, self.max_chars_size, num_cus, buf, 4096))
        return buf.value.decode()

    # -- the three derive_* of lib.rs:804-888 -----------------------------------------------------
    def derive_states(self, characters):
        ch = np.frombuffer(bytes(characters), dtype=np.uint8)
        n = len(ch)
        states = np.zeros((self.num_defs, n + 1), np.uint64)
        _check(lib.hrx_derive_states(self._need_ctx(), _ptr(ch, _u8p) if n else None, n, _ptr(states, _u64p)))
        return states

    def derive_substr_ids(self, states):
        states = _np(states, np.uint64)
        n = states.shape[1] - 1
        out = np.zeros((self.num_defs, n), np.uint64)
        _check(lib.hrx_derive_substr_ids(self._need_ctx(), _ptr(states, _u64p), n, _ptr(out, _u64p)))
        return out

    def derive_is_start_end(self, states, substr_ids):
        states = _np(states, np.uint64)
        substr_ids = _np(substr_ids, np.uint64)
        n = states.shape[1] - 1
        st = np.zeros((self.num_defs, n + 1), np.uint8)
        en = np.zeros((self.num_defs, n + 1), np.uint8)
        _check(lib.hrx_derive_is_start_end(self._need_ctx(), _ptr(states, _u64p), _ptr(substr_ids, _u64p), n,
                                           _ptr(st, _u8p), _ptr(en, _u8p)))
        return st.astype(bool), en.astype(bool)

    # -- match_substrs (lib.rs:311-773), integer columns ------------------------------------------
    def match_substrs(self, characters):
        ch = np.frombuffer(bytes(characters), dtype=np.uint8)
        n, M, D = len(ch), self.max_chars_size, self.num_defs
        cols = {k: np.zeros(M, np.uint64) for k in ("enable", "character", "masked_char", "masked_substr_id")}
        for k in ("state", "substr_id", "start_enable", "end_enable"):
            cols[k] = np.zeros((D, M), np.uint64)
        status = np.zeros(1, np.uint64)
        _check(lib.hrx_match_substrs(self._need_ctx(), _ptr(ch, _u8p) if n else None, n, M,
                                     *[_ptr(cols[k], _u64p) for k in ("enable", "character", "state", "substr_id",
                                                                       "start_enable", "end_enable", "masked_char",
                                                                       "masked_substr_id")], _ptr(status, _u64p)))
        return AssignedRegexResult(all_enable_flags=cols["enable"], all_characters=cols["character"],
                                   all_substr_ids=cols["masked_substr_id"], masked_characters=cols["masked_char"],
                                   states=cols["state"], substr_ids=cols["substr_id"],
                                   start_enables=cols["start_enable"], end_enables=cols["end_enable"],
                                   status=int(status[0]))

    # -- the batch surface (the build's addition; parity is per string with match_substrs) --------
    def witness_batch_host(self, chars2d, lens, out=None):
        """chars2d (B, stride) uint8 numpy, lens (B,) -> records (B,M,D) u32, masked (B,M) u16, status (B,) u64.
        out=(rec, msk, st): reuse the caller's arrays (fresh arrays cost a page fault per 4 KiB while the copy lands)."""
        chars2d = _np(chars2d, np.uint8)
        lens = _np(lens, np.uint32)
        B, stride = chars2d.shape
        M, D = self.max_chars_size, self.num_defs
        if out is None:
            out = np.zeros((B, M, D), np.uint32), np.zeros((B, M), np.uint16), np.zeros(B, np.uint64)
        rec, msk, st = out
        assert rec.shape == (B, M, D) and msk.shape == (B, M) and st.shape == (B,) and rec.flags.c_contiguous and msk.flags.c_contiguous
        _check(lib.hrx_witness_batch_host(self._need_ctx(), _ptr(chars2d, _u8p), stride, _ptr(lens, _u32p), B, M,
                                          _ptr(rec, _u32p), _ptr(msk, _u16p), _ptr(st, _u64p)))
        return rec, msk, st

    def alloc_outputs(self, B, device=None, pitched=False):
        """Device buffers for witness_batch: records int32 (B,M,D), masked int16 (B,M), status int64 (B,).
        pitched=True: the same shapes as views into buffers whose per-string pitch is hrx_recommended_pitches(M)
        (rows M.. of each string's slot are never written)."""
        dev = torch.device("cuda", self.device) if device is None else device
        M, D = self.max_chars_size, self.num_defs
        rp, mp = (recommended_pitches(M)[:2] if pitched else (M, M))
        st = torch.empty((B,), dtype=torch.int64, device=dev)
        if B * rp * D * 4 >= PLACED_FROM and dev.index in (None, self.device) and not torch.cuda.is_current_stream_capturing():     # several GB: a pair that does not collide
            pr, pmk = C.c_void_p(), C.c_void_p()
            _check(lib.hrx_alloc_output_pair(self._ctx, B * rp * D * 4, B * mp * 2, C.byref(pr), C.byref(pmk)))
            d = torch.device("cuda", self.device)
            rec = torch.as_tensor(_LibraryOwned(pr.value, B * rp * D * 4), device=d).view(torch.int32).view(B, rp, D)[:, :M]
            msk = torch.as_tensor(_LibraryOwned(pmk.value, B * mp * 2), device=d).view(torch.int16).view(B, mp)[:, :M]
            return rec, msk, st
        rec = torch.empty((B, rp, D), dtype=torch.int32, device=dev)[:, :M]
        msk = torch.empty((B, mp), dtype=torch.int16, device=dev)[:, :M]
        return rec, msk, st

    def alloc_outputs_position_major(self, B, device=None):
        """Flat device buffers for HRX_LAYOUT_POSITION_MAJOR: records int32 [ceil(M/4)*B*4*D], masked int16 [ceil(M/8)*B*8].
        (The status tensor — and, below the placement threshold, all three — comes from torch's caching allocator on the CURRENT stream: launch on that stream, or order
        the launch stream behind it, as with any torch tensor used on a side stream.)"""
        dev = torch.device("cuda", self.device) if device is None else device
        nr, nm = C.c_size_t(0), C.c_size_t(0)
        lib.hrx_position_major_sizes(B, self.max_chars_size, self.num_defs, C.byref(nr), C.byref(nm))
        st = torch.empty((B,), dtype=torch.int64, device=dev)
        if nr.value * 4 < PLACED_FROM or dev.index not in (None, self.device) or torch.cuda.is_current_stream_capturing():     # (the search allocates and measures: not inside a stream capture)
            return torch.empty((nr.value,), dtype=torch.int32, device=dev), torch.empty((nm.value,), dtype=torch.int16, device=dev), st
        # several GB: the library allocates the pair and keeps the masked-row buffer that does not collide with the records
        pr, pmk = C.c_void_p(), C.c_void_p()
        _check(lib.hrx_alloc_outputs_position_major(self._ctx, B, self.max_chars_size, C.byref(pr), C.byref(pmk)))
        d = torch.device("cuda", self.device)
        rec = torch.as_tensor(_LibraryOwned(pr.value, nr.value * 4), device=d).view(torch.int32)
        msk = torch.as_tensor(_LibraryOwned(pmk.value, nm.value * 2), device=d).view(torch.int16)
        return rec, msk, st

    def alloc_output_planes(self, B, device=None, stripes=None, chars=None, lens=None, chars_pm_stride=None):
        """hrx_alloc_output_planes: ([record buffers] int32, masked int16, status int64) — every def's records in a buffer of its own (one def: the ordinary records buffer, or with
        stripes=2 its two row stripes), each placed in a neighbourhood of the device memory of its own (include/hrx.h: the launch's write streams spread over the classes of the
        physical address space).  chars / lens (as witness_batch_planes takes them): hrx_alloc_output_planes_for_batch — the launches that choose among the candidate sets run this
        batch, not a stand-in."""
        dev = torch.device("cuda", self.device) if device is None else device
        D = self.num_defs
        R = 1 if stripes is None else int(stripes)
        assert R == 1 or D == 1
        n = D * R
        npl, nm = C.c_size_t(0), C.c_size_t(0)
        lib.hrx_position_major_stripe_sizes(B, self.max_chars_size, R, C.byref(npl), C.byref(nm))
        st = torch.empty((B,), dtype=torch.int64, device=dev)
        if npl.value * 4 * n < PLACED_FROM or dev.index not in (None, self.device) or torch.cuda.is_current_stream_capturing():
            return [torch.empty((npl.value,), dtype=torch.int32, device=dev) for _ in range(n)], torch.empty((nm.value,), dtype=torch.int16, device=dev), st
        # (one def, one buffer: the library's pool of record AND masked-row candidates + a dry launch from 1 GiB of records on, its measured arena pair below)
        arr, pmk = (C.c_void_p * n)(), C.c_void_p()
        if chars is not None:
            assert lens is not None and chars.is_cuda and lens.is_cuda and chars.dtype == torch.uint8 and lens.dtype == torch.int32 and lens.numel() == B
            layout = LAYOUT_POSITION_MAJOR
            if chars_pm_stride is None:
                assert chars.stride(1) == 1 and lens.is_contiguous() and chars.shape[0] == B
                stride = chars.stride(0)
            else:
                assert chars.is_contiguous() and chars.numel() == B * chars_pm_stride
                stride = int(chars_pm_stride)
                layout |= LAYOUT_INPUT_POSITION_MAJOR
            torch.cuda.current_stream(chars.device).synchronize()       # (the library's launches run on the context's stream)
            _check(lib.hrx_alloc_output_planes_for_batch(self._need_device(chars, lens), layout, chars.data_ptr(), stride, lens.data_ptr(), B, self.max_chars_size, n, arr, C.byref(pmk)))
        else:
            _check(lib.hrx_alloc_output_planes(self._ctx, B, self.max_chars_size, n, arr, C.byref(pmk)))
        d = torch.device("cuda", self.device)
        planes = [torch.as_tensor(_LibraryOwned(arr[k], npl.value * 4), device=d).view(torch.int32) for k in range(n)]
        msk = torch.as_tensor(_LibraryOwned(pmk.value, nm.value * 2), device=d).view(torch.int16)
        return planes, msk, st

    def witness_batch_planes(self, chars, lens, out=None, stream=None, chars_pm_stride=None):
        """hrx_witness_batch_device_planes: like witness_batch_position_major with the record planes in D buffers (out = (planes, masked, status) as alloc_output_planes gives them)."""
        assert chars.is_cuda and lens.is_cuda and chars.dtype == torch.uint8 and lens.dtype == torch.int32
        layout = LAYOUT_POSITION_MAJOR
        if chars_pm_stride is None:
            assert chars.stride(1) == 1 and lens.is_contiguous()
            B, stride = chars.shape[0], chars.stride(0)
        else:
            assert chars.is_contiguous() and chars.numel() == lens.numel() * chars_pm_stride
            B, stride = lens.numel(), int(chars_pm_stride)
            layout |= LAYOUT_INPUT_POSITION_MAJOR
        if out is None:
            out = self.alloc_output_planes(B, chars.device)
        planes, msk, st = out
        arr = (C.c_void_p * len(planes))(*[p.data_ptr() for p in planes])
        s = torch.cuda.current_stream(chars.device) if stream is None else stream
        _check(lib.hrx_witness_batch_device_planes(self._need_device(chars, lens, msk, st, *planes), layout, chars.data_ptr(), stride, lens.data_ptr(), B, self.max_chars_size,
                                                   arr, len(planes), msk.data_ptr(), st.data_ptr(), s.cuda_stream))
        return planes, msk, st

    def traffic_pass_planes(self, chars_pm, B, out, chars_pm_stride, stream=None):
        """hrx_traffic_pass_device_planes: the memory traffic of one record-planes launch over these buffers, no DFA work; OVERWRITES out."""
        planes, msk, _ = out
        arr = (C.c_void_p * len(planes))(*[p.data_ptr() for p in planes])
        s = torch.cuda.current_stream(chars_pm.device) if stream is None else stream
        _check(lib.hrx_traffic_pass_device_planes(self._need_device(chars_pm, msk, *planes), chars_pm.data_ptr(), int(chars_pm_stride), int(B), self.max_chars_size,
                                                  arr, len(planes), msk.data_ptr(), s.cuda_stream))

    def probe_write_pair(self, a, b, nbytes=None):
        """hrx_probe_write_pair: GB/s of two equal write streams over device tensors a and b (OVERWRITTEN): do they lie in colliding neighbourhoods of the device memory?"""
        nbytes = min(a.numel() * a.element_size(), b.numel() * b.element_size()) if nbytes is None else nbytes
        g = C.c_double(0.0)
        _check(lib.hrx_probe_write_pair(self._need_device(a, b), a.data_ptr(), b.data_ptr(), int(nbytes), C.byref(g)))
        return g.value

    def host_route_report(self):
        """hrx_ctx_host_route_report: what this config's last witness_batch_host call did (a dict)."""
        r = _HostRouteReportC()
        _check(lib.hrx_ctx_host_route_report(self._need_ctx(), C.byref(r), C.sizeof(r)))
        return {k: getattr(r, k) for k, _ in _HostRouteReportC._fields_}

    def set_option(self, option, value):
        """hrx_ctx_set_option (include/hrx.h HRX_OPT_*)."""
        _check(lib.hrx_ctx_set_option(self._need_ctx(), int(option), int(value)))

    def get_option(self, option):
        return lib.hrx_ctx_get_option(self._need_ctx(), int(option))

    def last_placement_report(self):
        """hrx_alloc_last_report: what the last placement-aware allocation of this config's context did (a dict)."""
        r = _PlaceReportC()
        _check(lib.hrx_alloc_last_report_sized(self._need_ctx(), C.byref(r), C.sizeof(r)))
        return {k: getattr(r, k) for k, _ in _PlaceReportC._fields_}

    def traffic_pass(self, chars_pm, B, out, chars_pm_stride, stream=None):
        """hrx_traffic_pass_device: the memory traffic of one position-major launch over these buffers, no DFA work; OVERWRITES out."""
        rec, msk, _ = out
        s = torch.cuda.current_stream(chars_pm.device) if stream is None else stream
        _check(lib.hrx_traffic_pass_device(self._need_device(chars_pm, rec, msk), chars_pm.data_ptr(), int(chars_pm_stride), int(B), self.max_chars_size,
                                           rec.data_ptr(), msk.data_ptr(), s.cuda_stream))

    def traffic_pass_string_major(self, chars, out, stream=None):
        """hrx_traffic_pass_device_layout(HRX_LAYOUT_STRING_MAJOR): the memory traffic of one string-major launch over these buffers (chars (B, stride) uint8;
        out = (records (B, M, D) view, masked (B, M) view, status) as alloc_outputs gives them, pitched or not), no DFA work; OVERWRITES out."""
        rec, msk, _ = out
        B, stride = chars.shape
        D = self.num_defs
        s = torch.cuda.current_stream(chars.device) if stream is None else stream
        _check(lib.hrx_traffic_pass_device_layout(self._need_device(chars, rec, msk), 0, chars.data_ptr(), int(stride), int(B), self.max_chars_size,
                                                  rec.data_ptr(), rec.stride(0) // D, msk.data_ptr(), msk.stride(0), s.cuda_stream))

    def chars_to_position_major_device(self, chars, out=None, stream=None):
        """hrx_chars_to_position_major_device: (B, stride) string-major bytes on the device (stride % 16 == 0, one contiguous string per row: the reference's
        input shape, lib.rs:311-315) -> the flat HRX_LAYOUT_INPUT_POSITION_MAJOR buffer (the library's own kernel; chars_to_position_major above does
        the same with torch ops).  Asynchronous on `stream` (default: torch's current stream)."""
        B, stride = chars.shape
        if not chars.is_contiguous():
            chars = chars.contiguous()
        if out is None:
            out = torch.empty((B * stride,), dtype=torch.uint8, device=chars.device)
        s = torch.cuda.current_stream(chars.device) if stream is None else stream
        _check(lib.hrx_chars_to_position_major_device(self._need_device(chars, out), chars.data_ptr(), int(stride), int(B), out.data_ptr(), s.cuda_stream))
        return out

    def witness_batch_position_major(self, chars, lens, out=None, stream=None, chars_pm_stride=None):
        """Like witness_batch, outputs in HRX_LAYOUT_POSITION_MAJOR (use position_major_to_string_major to view them per
        string).  chars: (B, stride) string-major, or — with chars_pm_stride=stride — the flat position-major buffer made by
        chars_to_position_major."""
        assert chars.is_cuda and lens.is_cuda and chars.dtype == torch.uint8 and lens.dtype == torch.int32
        layout = LAYOUT_POSITION_MAJOR
        if chars_pm_stride is None:
            assert chars.stride(1) == 1 and lens.is_contiguous()
            B, stride = chars.shape[0], chars.stride(0)
        else:
            assert chars.is_contiguous() and chars.numel() == lens.numel() * chars_pm_stride
            B, stride = lens.numel(), int(chars_pm_stride)
            layout |= LAYOUT_INPUT_POSITION_MAJOR
        if out is None:
            out = self.alloc_outputs_position_major(B, chars.device)
        rec, msk, st = out
        s = torch.cuda.current_stream(chars.device) if stream is None else stream
        _check(lib.hrx_witness_batch_device_layout(self._need_device(chars, lens, rec, msk, st), layout, chars.data_ptr(), stride,
                                                   lens.data_ptr(), B, self.max_chars_size, rec.data_ptr(), msk.data_ptr(),
                                                   st.data_ptr(), s.cuda_stream))
        return rec, msk, st

    def fr_columns(self, chars, lens, out, b_begin=0, b_count=None, position_major=False, chars_pm_stride=None, canonical=False,
                   stream=None):
        """expand the compact rows `out` = (records, masked, status) of a finished witness_batch* call into
        bn256::Fr cells for strings [b_begin, b_begin + b_count): int64 CUDA tensor [4 + 4 D][b_count][M][4] (limbs)."""
        rec, msk, _ = out
        B = lens.numel()
        b_count = B - b_begin if b_count is None else b_count
        M, D = self.max_chars_size, self.num_defs
        ncols = lib.hrx_fr_num_columns(D)
        cells = torch.empty((ncols, b_count, M, 4), dtype=torch.int64, device=lens.device)
        layout, rp, mp = LAYOUT_STRING_MAJOR, 0, 0
        if position_major:
            layout = LAYOUT_POSITION_MAJOR
            if chars_pm_stride is not None:
                layout |= LAYOUT_INPUT_POSITION_MAJOR
                stride = int(chars_pm_stride)
            else:
                stride = chars.stride(0)
        else:
            stride, rp, mp = chars.stride(0), rec.stride(0) // D, msk.stride(0)
        s = torch.cuda.current_stream(lens.device) if stream is None else stream
        if isinstance(rec, (list, tuple)):      # record planes (or the two row stripes of one def)
            arr = (C.c_void_p * len(rec))(*[p.data_ptr() for p in rec])
            _check(lib.hrx_fr_columns_device_planes(self._need_device(chars, lens, msk, cells, *rec), layout, chars.data_ptr(), stride, lens.data_ptr(), arr, len(rec), msk.data_ptr(),
                                                    B, M, b_begin, b_count, cells.data_ptr(), FR_CANONICAL if canonical else 0, s.cuda_stream))
            return cells
        _check(lib.hrx_fr_columns_device(self._need_device(chars, lens, rec, msk, cells), layout, chars.data_ptr(), stride, lens.data_ptr(), rec.data_ptr(), rp,
                                         msk.data_ptr(), mp, B, M, b_begin, b_count, cells.data_ptr(),
                                         FR_CANONICAL if canonical else 0, s.cuda_stream))
        return cells

    def witness_batch(self, chars, lens, out=None, stream=None):
        """Device-resident batch: chars (B, stride) uint8 CUDA tensor (stride % 16 == 0), lens (B,) int32 CUDA tensor.
        Asynchronous on `stream` (default: torch's current stream).  Returns (records, masked, status) tensors whose
        bit patterns are the u32/u16/u64 layouts of include/hrx.h."""
        assert chars.is_cuda and lens.is_cuda and chars.dtype == torch.uint8 and lens.dtype == torch.int32
        assert chars.stride(1) == 1 and lens.is_contiguous()
        B, stride = chars.shape[0], chars.stride(0)       # a (B, n) view of a wider buffer is fine
        if out is None:
            out = self.alloc_outputs(B, chars.device)
        rec, msk, st = out
        D = self.num_defs
        assert rec.stride(2) == 1 and rec.stride(1) == D and rec.stride(0) % D == 0 and msk.stride(1) == 1
        s = torch.cuda.current_stream(chars.device) if stream is None else stream
        _check(lib.hrx_witness_batch_device_pitched(self._need_device(chars, lens, rec, msk, st), chars.data_ptr(), stride, lens.data_ptr(), B,
                                                    self.max_chars_size, rec.data_ptr(), rec.stride(0) // D,
                                                    msk.data_ptr(), msk.stride(0), st.data_ptr(), s.cuda_stream))
        return rec, msk, st


FR_CANONICAL = 1                       # HRX_FR_CANONICAL
FR_COLUMN_NAMES = lambda D: (["char_enable", "characters"] + [n % d for d in range(D) for n in ("states[%d]", "substr_ids[%d]", "start_enable[%d]", "end_enable[%d]")]
                             + ["masked_characters", "all_substr_ids"])


def fr_from_u64(v, canonical=False):
    """F::from(v) for F = bn256::Fr as the library computes it: 4 little-endian u64 limbs (Montgomery form unless canonical)."""
    out = (C.c_uint64 * 4)()
    lib.hrx_fr_from_u64(int(v), FR_CANONICAL if canonical else 0, out)
    return [int(x) for x in out]


class MultiDevice:
    """hrx_multi_*: one batch of host strings sharded by index over several devices (or several shards of one device), no
    collective.  MultiDevice(config, [0, 1, 2, 3]).witness_batch_host(chars, lens) == config.witness_batch_host(chars, lens)."""

    def __init__(self, config, devices):
        self._cfg = config
        self._h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        _check(lib.hrx_multi_create(config._defs.h, arr, len(devices), C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib.hrx_multi_destroy(self._h)
            self._h = None

    @property
    def num_shards(self):
        return lib.hrx_multi_num_shards(self._h)

    def shard_device(self, shard):
        return lib.hrx_multi_shard_device(self._h, shard)

    def witness_batch_device(self, shards, layout=LAYOUT_POSITION_MAJOR | LAYOUT_INPUT_POSITION_MAJOR, chars_stride=None):
        """hrx_multi_witness_batch_device: `shards` = one (chars, lens, (records, masked, status)) per shard, CUDA tensors on
        that shard's device (outputs as made by alloc_outputs_position_major / alloc_outputs on that device).  Asynchronous:
        one kernel per shard on the shard's own stream; synchronize() waits for all.  No PCIe traffic, no collective.
        Every shard's stream first waits for what torch's current stream of that device has been given so far (the kernels
        that produced the inputs).  chars_stride: the per-string capacity in bytes; required for flat position-major
        inputs unless it follows from the buffer sizes (numel(chars) / numel(lens), the same for every shard)."""
        n = self.num_shards
        assert len(shards) == n
        vpa, sza = C.c_void_p * n, C.c_size_t * n
        chars, lens, recs, msks, sts, counts = vpa(), vpa(), vpa(), vpa(), vpa(), sza()
        strides = set()
        for r, (c, l, (rec, msk, st)) in enumerate(shards):
            for t in (c, l, rec, msk, st):
                if t.numel() and t.device.index != self.shard_device(r):
                    raise HrxError(HRX_ERR_ARG, "shard %d: tensor on %s, shard lives on cuda:%d" % (r, t.device, self.shard_device(r)))
            chars[r], lens[r], recs[r], msks[r], sts[r], counts[r] = c.data_ptr(), l.data_ptr(), rec.data_ptr(), msk.data_ptr(), st.data_ptr(), l.numel()
            if chars_stride is None and l.numel():
                if c.dim() == 2 and not (layout & LAYOUT_INPUT_POSITION_MAJOR):
                    strides.add(int(c.stride(0)))
                elif c.numel() % l.numel() == 0:
                    strides.add(c.numel() // l.numel())
                else:
                    raise HrxError(HRX_ERR_ARG, "shard %d: chars_stride not given and numel(chars) is not a multiple of numel(lens)" % r)
        if chars_stride is not None:
            stride = int(chars_stride)
        elif len(strides) == 1:
            stride = strides.pop()
        elif not strides:
            stride = 16      # every shard is empty: nothing is launched
        else:
            raise HrxError(HRX_ERR_ARG, "chars_stride not given and the shards' buffers imply different strides: %s" % sorted(strides))
        if stride % 16 or stride < 16:
            raise HrxError(HRX_ERR_ARG, "chars stride %d: must be a multiple of 16 bytes, >= 16" % stride)
        for r in range(n):     # order each shard's private stream behind the producer of its inputs
            if counts[r]:
                d = self.shard_device(r)
                torch.cuda.ExternalStream(lib.hrx_multi_shard_stream(self._h, r), device=torch.device("cuda", d)).wait_stream(torch.cuda.current_stream(d))
        _check(lib.hrx_multi_witness_batch_device(self._h, layout, chars, stride, lens, counts, self._cfg.max_chars_size, recs, msks, sts))

    def synchronize(self):
        _check(lib.hrx_multi_synchronize(self._h))

    def witness_batch_host(self, chars2d, lens, out=None):
        chars2d = _np(chars2d, np.uint8)
        lens = _np(lens, np.uint32)
        B, stride = chars2d.shape
        M, D = self._cfg.max_chars_size, self._cfg.num_defs
        if out is None:
            out = np.zeros((B, M, D), np.uint32), np.zeros((B, M), np.uint16), np.zeros(B, np.uint64)
        rec, msk, st = out
        _check(lib.hrx_multi_witness_batch_host(self._h, _ptr(chars2d, _u8p), stride, _ptr(lens, _u32p), B, M,
                                                _ptr(rec, _u32p), _ptr(msk, _u16p), _ptr(st, _u64p)))
        return rec, msk, st


def decode_status(s):
    s = int(s) & 0xFFFFFFFFFFFFFFFF
    code = s & 0xff
    if code == 0:
        return {"code": 0, "accept": (s >> 8) & 0xffffffff}
    if code == 1:
        return {"code": 1, "def": (s >> 8) & 0xff, "char": (s >> 16) & 0xff, "state": (s >> 24) & 0xffff, "pos": s >> 40}
    if code == 2:
        return {"code": 2, "pos": s >> 40}
    return {"code": code}
