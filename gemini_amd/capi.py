"""ctypes loader for libgemini_hip.so (the C ABI of include/gemini_hip.h).

Fails loudly: there is no CPU or PyTorch fallback for any entry point.  If the shared library is
missing (run `python -c "import __graft_entry__ as g; g.build()"`) or no GPU is visible, callers
get an exception, never a silently different code path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GM_LIB_PATH: development hook (tools/ A/B runs of differently built libraries); the product path is the in-tree .so
LIB_PATH = os.environ.get("GM_LIB_PATH") or os.path.join(_HERE, "libgemini_hip.so")

GM_OK = 0
ERRORS = {-1: "GM_EINVAL", -2: "GM_ENOTINIT", -3: "GM_EHANDLE", -4: "GM_EHIP", -5: "GM_ENOMEM", -6: "GM_ESTATE"}

# every symbol include/gemini_hip.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "gm_init", "gm_shutdown", "gm_last_error", "gm_abi_version",
    "gm_g1_msm", "gm_g1_bases_register", "gm_g1_bases_free", "gm_g1_bases_len", "gm_g1_bases_download", "gm_g1_bases_precompute", "gm_set_auto_tables", "gm_g1_bases_table_info", "gm_pool_trim", "gm_g1_release_spare_tables", "gm_mem_stats", "gm_mem_reset_peak", "gm_runtime_info", "gm_snark_footprint", "gm_psnark_footprint", "gm_footprint_admit",
    "gm_g1_msm_h", "gm_g1_msm_v", "gm_g1_msm_v_batch", "gm_g1_msm_v_batch_partial", "gm_g1_msm_v_batch_at", "gm_g1_msm_d", "gm_g1_msm_d_partial", "gm_g1_sum",
    "gm_g1_msm_stream_new", "gm_g1_msm_stream_new_h", "gm_g1_msm_stream_add", "gm_g1_msm_stream_finalize", "gm_g1_msm_stream_free", "gm_host_alloc", "gm_host_free",
    "gm_g1_fixed_base_register", "gm_g1_srs_register", "gm_g1_srs_register_segments", "gm_set_msm_window", "gm_set_msm_table_min", "gm_set_msm_affine_levels", "gm_set_msm_split", "gm_set_msm_glv", "gm_prof_enable", "gm_prof_read", "gm_prof_read_clock",
    "gm_idx_register", "gm_idx_free", "gm_fr_gather", "gm_fr_alg_hash", "gm_fr_plookup_set", "gm_fr_add_scalar", "gm_fr_shift_monic",
    "gm_fr_acc_product", "gm_fr_tensor_range", "gm_fr_tensor_gather", "gm_fr_powers_gather", "gm_fr_alg_hash_from", "gm_fr_powers_range", "gm_fr_plookup_set_block", "gm_fr_shift_block", "gm_fr_product", "gm_fr_acc_product_block",
    "gm_sc_set_shard_rounds", "gm_sc_round_begin_many", "gm_psnark_shard_block", "gm_psnark_shard_level", "gm_psnark_shard_footprint", "gm_psnark_shard_key_new", "gm_psnark_index_sharded", "gm_psnark_new_time_sharded", "gm_snark_new_elastic_sharded",
    "gm_fr_vec_alloc", "gm_fr_vec_free", "gm_fr_vec_len", "gm_fr_vec_upload", "gm_fr_vec_download",
    "gm_fr_vec_fill", "gm_fr_vec_ptr", "gm_fr_vec_set_len",
    "gm_fr_reverse", "gm_fr_stride", "gm_fr_fold", "gm_fr_fold_chain", "gm_fr_powers", "gm_fr_tensor", "gm_fr_hadamard", "gm_fr_ip", "gm_fr_eval_le", "gm_fr_eval_le_batch",
    "gm_fr_lincomb", "gm_fr_scale_into", "gm_fr_scale_into_many", "gm_fr_add_at", "gm_fr_div_vanishing",
    "gm_spm_register", "gm_spm_free", "gm_spm_mul", "gm_spm_shape", "gm_snark_new_time", "gm_snark_new_elastic", "gm_psnark_new_time", "gm_psnark_new_elastic", "gm_psnark_preprocess", "gm_psnark_preprocess_free", "gm_psnark_index", "gm_spm_download",
    "gm_sc_new", "gm_sc_new_v", "gm_sc_new_borrow", "gm_sc_round", "gm_sc_round_begin", "gm_sc_round_end", "gm_sc_fold", "gm_sc_rounds", "gm_sc_final", "gm_sc_free",
    "gm_sc_set_shard", "gm_sc_lens", "gm_sc_download",
    "gm_sp_new", "gm_sp_new_v", "gm_sp_new_borrow", "gm_sp_round", "gm_sp_fold", "gm_sp_rounds", "gm_sp_final", "gm_sp_to_time", "gm_sp_free",
    "gm_sc_set_herring", "gm_hg1_new", "gm_hg1_round", "gm_hg1_fold", "gm_hg1_rounds", "gm_hg1_final", "gm_hg1_free",
    "gm_transcript_new", "gm_transcript_free", "gm_transcript_append_message", "gm_transcript_challenge_bytes",
    "gm_transcript_append_fr", "gm_transcript_append_g1", "gm_transcript_set_g1_encoding", "gm_transcript_challenge_fr", "gm_sumcheck_prove", "gm_sumcheck_prove_batch",
    "gm_dist_init_hook", "gm_dist_rccl_unique_id", "gm_dist_init_rccl", "gm_dist_init_shm", "gm_dist_init_rccl_node", "gm_dist_finalize", "gm_dist_abort", "gm_dist_info", "gm_dist_allgather_host",
    "gm_dist_allgather_vec", "gm_dist_stats", "gm_dist_selftest", "gm_dist_allgather_host_class", "gm_dist_stats_routes", "gm_dist_bench", "gm_dist_reblock_vecs",
    "gm_g1_bases_set_cyclic", "gm_g1_srs_register_cyclic", "gm_ck_len", "gm_ck_msm", "gm_ck_msm_batch", "gm_sumcheck_prove_sharded",
    "gm_snark_shard_key_new", "gm_snark_new_time_sharded",
]


class GeminiHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{ERRORS.get(code, code)}: {msg}")
        self.code = code


_lib = None


def _load_torch_runtime_first():
    """The PyTorch-ROCm wheel ships its own libamdhip64.so / libhsa-runtime64.so (ROCm 7.0 here) while this library links
    the system ones (/opt/rocm, 7.2, different sonames): both HIP runtimes end up in the process.  That works -- device
    pointers of either are valid for the kernels of the other -- as long as torch's libraries are mapped FIRST; the other
    way round torch binds to the system HSA runtime and reports "No HIP GPUs are available" (tools/torch_order_probe.py).
    Python callers of this package use torch for device memory and torch.distributed, so import it before the dlopen.
    C / C++ / Rust embedders of the C ABI are not concerned.  GM_NO_TORCH_PRELOAD=1 skips this."""
    if os.environ.get("GM_NO_TORCH_PRELOAD") == "1":
        return
    try:
        import torch  # noqa: F401
    except Exception:  # torch is optional for the C ABI itself
        pass


def load() -> C.CDLL:
    """dlopen the library (no GPU needed for this step)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(python -c 'import __graft_entry__ as g; g.build()'). There is no fallback path."
            )
        _load_torch_runtime_first()
        lib = C.CDLL(LIB_PATH)
        lib.gm_last_error.restype = C.c_char_p
        _lib = lib
    return _lib


def check(rc: int):
    if rc != GM_OK:
        raise GeminiHipError(rc, load().gm_last_error().decode())


_initialised = False


def init(device: int | None = None):
    """gm_init on LOCAL_RANK (one process per GPU)."""
    global _initialised
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    check(load().gm_init(C.c_int(device)))
    _initialised = True


def ensure_init():
    if not _initialised:
        init()


MEM_FIELDS = ("device_total", "device_free", "held", "held_peak", "pool_cached", "in_use", "in_use_peak", "tables", "keys",
              "spare_table_releases", "msm_workspaces", "reserved")


def mem_stats() -> dict:
    """gm_mem_stats: the library's device-memory bookkeeping (bytes; include/gemini_hip.h)."""
    out = np.zeros(12, dtype=np.uint64)
    check(load().gm_mem_stats(ptr(out)))
    return {k: int(v) for k, v in zip(MEM_FIELDS, out)}


FOOTPRINT_FIELDS = ("vectors", "workspaces_to_grow", "needed", "available")


def snark_footprint(ck_handle: int, num_constraints: int, elastic: bool = False) -> dict:
    out = np.zeros(4, dtype=np.uint64)
    check(load().gm_snark_footprint(C.c_uint64(ck_handle), C.c_size_t(num_constraints), C.c_int(int(elastic)), ptr(out)))
    return {k: int(v) for k, v in zip(FOOTPRINT_FIELDS, out)}


def psnark_footprint(ck_handle: int, num_variables: int, nnz: int, elastic: int = 0) -> dict:
    """elastic: 0 time prover, 1 elastic prover in the resident schedule, 2 literal (min_device_chunk = 1)"""
    out = np.zeros(4, dtype=np.uint64)
    check(load().gm_psnark_footprint(C.c_uint64(ck_handle), C.c_size_t(num_variables), C.c_size_t(nnz), C.c_int(int(elastic)), ptr(out)))
    return {k: int(v) for k, v in zip(FOOTPRINT_FIELDS, out)}


def psnark_shard_footprint(key_handle: int, num_constraints: int, num_variables: int, nnz: int, block: int, world: int) -> dict:
    """one rank of gm_psnark_new_time_sharded (z and the instance's blocks are the caller's)"""
    out = np.zeros(4, dtype=np.uint64)
    check(load().gm_psnark_shard_footprint(C.c_uint64(key_handle), C.c_size_t(num_constraints), C.c_size_t(num_variables), C.c_size_t(nnz), C.c_size_t(block),
                                           C.c_int(world), C.c_int(0), ptr(out)))
    return {k: int(v) for k, v in zip(FOOTPRINT_FIELDS, out)}


def runtime_info() -> dict:
    out = (C.c_int * 4)()
    check(load().gm_runtime_info(out))
    return {"compute_units": out[0], "batch_tail_cus": out[1], "zero_copy": out[2], "small_lanes": out[3]}


def mem_reset_peak():
    check(load().gm_mem_reset_peak())


def ptr(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def u64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)
