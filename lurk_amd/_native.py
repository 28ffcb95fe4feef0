"""ctypes binding of the lurkhip C ABI (include/lurkhip.h).

The HIP library is the product: there is no Python or CPU fallback.  If
``liblurkhip.so`` has not been built this module raises at import time, and
every compute call fails with ``LurkHipError`` when no HIP device is usable.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LURKHIP_LIB_PATH selects another build of the same ABI (the A/B variant liblurkhip_declared.so of tests/test_mad_ab_gpu.py)
LIB_PATH = os.environ.get("LURKHIP_LIB_PATH") or os.path.join(_HERE, "liblurkhip.so")


class LurkHipError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"lurkhip status {status}: {message}")
        self.status = status
        self.message = message


OK = 0
ERR_INVALID_ARG = -1
ERR_NO_DEVICE = -2
ERR_HIP = -3
ERR_OOM = -4
ERR_UNSUPPORTED = -5
ERR_EXEC = -6
ERR_PARSE = -7

REPR_CANONICAL = 0
REPR_MONTY = 1

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C lurk_amd/csrc`). lurk_amd has no CPU fallback."
    )

# One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 and a process that maps
# both that copy and /opt/rocm's ends up with two runtimes, the second of which sees no devices.  When
# torch is installed, load its runtime first so liblurkhip's DT_NEEDED libamdhip64.so.7 resolves to it.
try:  # pragma: no cover - depends on the environment
    import torch  # noqa: F401
except Exception:  # torch is plumbing for device memory / torch.distributed, not a requirement
    torch = None

lib = C.CDLL(LIB_PATH)

_p = C.c_void_p
_u32p = C.c_void_p  # raw addresses (numpy .ctypes.data, torch .data_ptr())
_sz = C.c_size_t
_i32 = C.c_int32
_i64 = C.c_int64

# name -> (restype, argtypes); kept in the order of include/lurkhip.h
SIGNATURES = {
    "lurkhip_abi_version": (_i32, []),
    "lurkhip_ctx_create": (_i32, [_i32, C.POINTER(_p)]),
    "lurkhip_ctx_create_with_priority": (_i32, [_i32, _i32, C.POINTER(_p)]),
    "lurkhip_ctx_create_on_stream": (_i32, [_i32, _p, C.POINTER(_p)]),
    "lurkhip_ctx_create_beside": (_i32, [_p, C.POINTER(_p)]),
    "lurkhip_ctx_overlap_probe": (_i32, [_p, _p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "lurkhip_ctx_destroy": (_i32, [_p]),
    "lurkhip_ctx_sync": (_i32, [_p]),
    "lurkhip_event_record": (_i32, [_p, C.POINTER(_p)]),
    "lurkhip_event_wait": (_i32, [_p, _p]),
    "lurkhip_event_destroy": (_i32, [_p]),
    "lurkhip_last_error": (C.c_char_p, [_p]),
    "lurkhip_malloc": (_i32, [_p, _sz, C.POINTER(_p)]),
    "lurkhip_free": (_i32, [_p, _p]),
    "lurkhip_memcpy_h2d": (_i32, [_p, _p, _p, _sz]),
    "lurkhip_memcpy_d2h": (_i32, [_p, _p, _p, _sz]),
    "lurkhip_timer_start": (_i32, [_p]),
    "lurkhip_timer_stop": (_i32, [_p, C.POINTER(C.c_float)]),
    "lurkhip_profile_enable": (_i32, [_p, _i32]),
    "lurkhip_profile_span_begin": (_i32, [_p, C.c_char_p]),
    "lurkhip_profile_span_end": (_i32, [_p, C.c_char_p]),
    "lurkhip_profile_reset": (_i32, [_p]),
    "lurkhip_profile_read": (_i32, [_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "lurkhip_perm16": (_i32, [_p, _sz, _p, _p, _i32]),
    "lurkhip_perm16_dev": (_i32, [_p, _sz, _p, _p, _i32]),
    "lurkhip_pool_trim": (_i32, [_p]),
    "lurkhip_pool_stats": (_i32, [_p, _p]),
    "lurkhip_pool_reset_peak": (_i32, [_p]),
    "lurkhip_debug_inject_alloc_failures": (_i32, [_p, _i32]),
    "lurkhip_poseidon2_num_cols": (_i32, [_i32]),
    "lurkhip_poseidon2_permute": (_i32, [_p, _i32, _sz, _u32p, _u32p, _i32]),
    "lurkhip_poseidon2_permute_dev": (_i32, [_p, _i32, _sz, _u32p, _u32p, _i32]),
    "lurkhip_poseidon2_hash8": (_i32, [_p, _i32, _sz, _u32p, _u32p, _i32]),
    "lurkhip_poseidon2_hash8_dev": (_i32, [_p, _i32, _sz, _u32p, _u32p, _i32]),
    "lurkhip_poseidon2_wide_witness": (_i32, [_p, _i32, _sz, _u32p, _u32p, _i32]),
    "lurkhip_poseidon2_wide_witness_dev": (_i32, [_p, _i32, _sz, _u32p, _u32p, _i32]),
    "lurkhip_poseidon2_trace_shape": (_i32, [_i32, _sz, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "lurkhip_poseidon2_trace": (_i32, [_p, _i32, _sz, _u32p, _u32p, _i32]),
    "lurkhip_poseidon2_trace_dev": (_i32, [_p, _i32, _sz, _u32p, _u32p, _i32]),
    "lurkhip_set_merkle_poseidon2": (_i32, [_p, _i32, _u32p, _u32p, _u32p]),
    "lurkhip_protocol_profile_preset": (_i32, [C.c_char_p, _p]),
    "lurkhip_set_protocol_profile": (_i32, [_p, _p]),
    "lurkhip_get_protocol_profile": (_i32, [_p, _p]),
    "lurkhip_coset_lde": (_i32, [_p, _i32, _i32, _i32, _u32p, _u32p, _i32]),
    "lurkhip_coset_lde_dev": (_i32, [_p, _i32, _i32, _i32, _u32p, _u32p, _i32]),
    "lurkhip_mmcs_commit": (_i32, [_p, _i32, _p, _u32p, _u32p, _i32, C.POINTER(_p), _u32p]),
    "lurkhip_commit": (_i32, [_p, _i32, _p, _u32p, _u32p, _i32, _i32, _i32, C.POINTER(_p), _u32p]),
    "lurkhip_commit_dev": (_i32, [_p, _i32, _p, _u32p, _u32p, _i32, _i32, _i32, C.POINTER(_p), _u32p]),
    "lurkhip_commit_dev_sparse": (_i32, [_p, _i32, _p, _u32p, _u32p, _i32, _i32, _i32, C.POINTER(_p), _u32p, _u32p]),
    "lurkhip_commit_cosets_dev": (_i32, [_p, _i32, _p, _u32p, _u32p, _u32p, _i32, _i32, C.POINTER(_p), _u32p]),
    "lurkhip_commitment_free": (_i32, [_p, _p]),
    "lurkhip_commitment_root": (_i32, [_p, _p, _u32p, _i32]),
    "lurkhip_commitment_matrix_dev": (_i32, [_p, _p, _i32, C.POINTER(_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "lurkhip_commitment_matrix_pitch": (_i32, [_p, _p, _i32, C.POINTER(C.c_uint32)]),
    "lurkhip_commitment_open": (_i32, [_p, _p, C.c_uint64, _u32p, _u32p, _i32]),
    "lurkhip_trace_func_dev": (_i32, [_p, _u32p, _u32p, C.c_uint32, C.c_uint32, C.c_uint32, _u32p, _u32p, _u32p, _u32p, _p, _u32p, _u32p, _i32]),
    "lurkhip_trace_mem_dev": (_i32, [_p, C.c_uint32, C.c_uint32, C.c_uint32, _u32p, _u32p, _u32p, _i32]),
    "lurkhip_trace_bytes_dev": (_i32, [_p, _u32p, _i32, _u32p, _i32]),
    "lurkhip_trace_bytes_preprocessed_dev": (_i32, [_p, _u32p, _i32]),
    "lurkhip_zstore_new": (_i32, [_p, C.POINTER(_p)]),
    "lurkhip_zstore_free": (_i32, [_p]),
    "lurkhip_zstore_last_error": (C.c_char_p, [_p]),
    "lurkhip_zstore_intern_dag": (_i32, [_p, C.c_uint32, _p, _p]),
    "lurkhip_zstore_stats": (_i32, [_p, _p]),
    "lurkhip_zstore_set_inverse_tables": (_i32, [_p, C.c_uint64, _p, C.c_uint64, _p]),
    "lurkhip_zstore_memoize_dag": (_i32, [_p, C.c_uint32, _p]),
    "lurkhip_zstore_fetch": (_i32, [_p, _p, _p]),
    "lurkhip_zstore_dag_export": (C.c_int64, [_p, C.c_uint32, _p, _p, C.c_uint64]),
    "lurkhip_toplevel_new": (_i32, [C.c_char_p, _i32, C.POINTER(_p)]),
    "lurkhip_toplevel_from_bytecode": (_i32, [_p, C.c_uint64, C.POINTER(_p)]),
    "lurkhip_toplevel_to_bytecode": (C.c_int64, [_p, _p, C.c_uint64]),
    "lurkhip_toplevel_free": (_i32, [_p]),
    "lurkhip_toplevel_num_funcs": (_i32, [_p]),
    "lurkhip_toplevel_func_index": (_i32, [_p, C.c_char_p]),
    "lurkhip_toplevel_func_info": (_i32, [_p, _i32, _u32p]),
    "lurkhip_lair_last_error": (C.c_char_p, []),
    "lurkhip_record_new": (_i32, [_p, C.POINTER(_p)]),
    "lurkhip_record_free": (_i32, [_p]),
    "lurkhip_record_clean": (_i32, [_p]),
    "lurkhip_execute": (_i32, [_p, _i32, _u32p, C.c_uint32, _u32p]),
    "lurkhip_record_inject_inv_query": (_i32, [_p, _i32, _u32p, C.c_uint32, _u32p, C.c_uint32]),
    "lurkhip_record_count": (_i64, [_p, _i32, _i32]),
    "lurkhip_record_public_values": (_i32, [_p, _u32p]),
    "lurkhip_record_num_shards": (_i64, [_p, C.c_uint32]),
    "lurkhip_func_trace_shape": (_i32, [_p, _i32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "lurkhip_generate_trace_func": (_i32, [_p, _p, _p, _i32, C.c_uint32, C.c_uint32, _u32p, _i32]),
    "lurkhip_generate_trace_func_dev": (_i32, [_p, _p, _p, _i32, C.c_uint32, C.c_uint32, _u32p, _i32]),
    "lurkhip_trace_compile": (_i32, [_p, _p, _i32]),
    "lurkhip_trace_compile_check": (_i32, [_p, _i32, _p, C.c_uint32]),
    "lurkhip_trace_source": (_i32, [_p, _i32, _p, C.c_uint32]),
    "lurkhip_func_trace_prepare": (_i32, [_p, _p, _p, _i32, C.c_uint32, C.c_uint32, C.POINTER(_p)]),
    "lurkhip_func_trace_prepare_many": (_i32, [_p, _p, _p, C.c_uint32, C.POINTER(_i32), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_p)]),
    "lurkhip_mem_trace_prepare": (_i32, [_p, _p, C.c_uint32, C.POINTER(_p)]),
    "lurkhip_bytes_trace_prepare": (_i32, [_p, _p, C.c_uint32, C.POINTER(_p)]),
    "lurkhip_func_trace_shape_of": (_i32, [_p, C.POINTER(C.c_uint64)]),
    "lurkhip_func_trace_run": (_i32, [_p, _p, _u32p, _i32]),
    "lurkhip_func_trace_run_many": (_i32, [_p, C.c_uint32, C.POINTER(_p), C.POINTER(_p), _i32]),
    "lurkhip_trace_group_layout": (_i32, [C.c_uint32, _u32p, _u32p, _u32p, _u32p, _u32p, _u32p]),
    "lurkhip_func_trace_run_pitched": (_i32, [_p, _p, _u32p, C.c_uint32, _i32]),
    "lurkhip_func_trace_export_size": (_i32, [_p, C.POINTER(C.c_uint64)]),
    "lurkhip_func_trace_export": (_i32, [_p, _p, _u32p, C.c_uint64]),
    "lurkhip_func_trace_import": (_i32, [_p, _u32p, C.c_uint64, C.POINTER(_p)]),
    "lurkhip_func_trace_run_many_pitched": (_i32, [_p, C.c_uint32, C.POINTER(_p), C.POINTER(_p), _u32p, _i32]),
    "lurkhip_func_trace_free": (_i32, [_p, _p]),
    "lurkhip_mem_trace_shape": (_i32, [_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "lurkhip_generate_trace_mem": (_i32, [_p, _p, C.c_uint32, _u32p, _i32]),
    "lurkhip_generate_trace_bytes": (_i32, [_p, _p, C.c_uint32, _u32p, _i32]),
    "lurkhip_air_func": (_i32, [_p, _i32, C.POINTER(_p)]),
    "lurkhip_air_mem": (_i32, [C.c_uint32, C.POINTER(_p)]),
    "lurkhip_air_bytes": (_i32, [C.POINTER(_p)]),
    "lurkhip_air_entrypoint": (_i32, [C.c_uint32, C.c_uint32, C.POINTER(_p)]),
    "lurkhip_air_poseidon2": (_i32, [_i32, C.POINTER(_p)]),
    "lurkhip_air_free": (_i32, [_p]),
    "lurkhip_air_name": (C.c_char_p, [_p]),
    "lurkhip_air_info": (_i32, [_p, _u32p]),
    "lurkhip_air_interaction_sizes": (_i32, [_p, _u32p, C.c_uint32]),
    "lurkhip_air_program": (_i32, [_p, _i32, C.c_uint32, _u32p, C.c_uint32]),
    "lurkhip_air_compile": (_i32, [_p, _p]),
    "lurkhip_air_compile_check": (_i32, [_p, _p, C.c_uint32]),
    "lurkhip_air_eval_rows": (_i32, [_p, _p, C.c_uint32, _u32p, _u32p, _u32p, _u32p, _u32p, _u32p, _u32p, _u32p]),
    "lurkhip_air_check_trace_dev": (_i32, [_p, _p, C.c_uint32, _u32p, _u32p, _u32p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "lurkhip_permutation_trace_dev": (_i32, [_p, _p, C.c_uint32, _u32p, _u32p, _u32p, _u32p, _u32p]),
    "lurkhip_challenger_new": (_i32, [_p, C.POINTER(_p)]),
    "lurkhip_challenger_clone": (_i32, [_p, C.POINTER(_p)]),
    "lurkhip_challenger_free": (_i32, [_p]),
    "lurkhip_challenger_observe": (_i32, [_p, _u32p, C.c_uint32]),
    "lurkhip_challenger_sample": (_i32, [_p, _u32p, C.c_uint32]),
    "lurkhip_challenger_sample_bits": (_i32, [_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "lurkhip_setup": (_i32, [_p, _i32, _p, _u32p, _u32p, _i32, C.POINTER(_p), _u32p]),
    "lurkhip_pk_free": (_i32, [_p, _p]),
    "lurkhip_shard_commit": (_i32, [_p, _i32, _p, _u32p, _p, _p, _i32, C.POINTER(_p), _u32p]),
    "lurkhip_shard_commit_pitched": (_i32, [_p, _i32, _p, _u32p, _p, _u32p, _p, _i32, C.POINTER(_p), _u32p]),
    "lurkhip_shard_free": (_i32, [_p, _p]),
    "lurkhip_shard_prove": (_i32, [_p, _p, _p, _p, _u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_p)]),
    "lurkhip_open": (_i32, [_p, _i32, _p, _p, _p, _p, C.c_uint32, C.c_uint32, C.POINTER(_p)]),
    "lurkhip_proof_words": (_i64, [_p]),
    "lurkhip_crypto_proof_verify": (_i32, [_p, _p, _p, C.c_uint32, _u32p, _u32p, _u32p, C.c_uint32, _p, C.c_uint64, _u32p, C.c_uint32, C.c_uint32,
                                            C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint32]),
    "lurkhip_cached_proof_verify": (_i32, [_p, _p, _p, C.c_uint32, _u32p, _u32p, _u32p, C.c_uint32, _p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                            _u32p, C.c_char_p, C.c_uint32]),
    "lurkhip_machine_verify": (_i32, [_p, _p, C.c_uint32, _u32p, _u32p, _u32p, C.c_uint32, _p, _p, C.c_uint32, C.c_char_p, C.c_uint32]),
    "lurkhip_logup_multiplicities": (_i32, [_p, C.c_uint32, C.c_uint32, _p, _p, _p, _p, _p]),
    "lurkhip_logup_permutation_trace": (_i32, [_p, C.c_uint32, C.c_uint32, C.c_uint32, _p, _p, _p, _p, _p, C.c_uint64, _p, _p, _p, _i32, _p, _p]),
    "lurkhip_logup_eval_constraints": (_i32, [_p, C.c_uint32, C.c_uint32, C.c_uint32, _p, _p, _p, _p, _p, _p, _p, C.c_uint64, _p, _p, _p, _p, _p, _i32, _p]),
    "lurkhip_shard_proof_bincode": (_i64, [_p, C.c_uint64, _i32, _p, _i32, _p, C.c_uint64]),
    "lurkhip_crypto_proof_bincode": (_i64, [_i32, _p, _p, _i32, _p, C.c_char_p, _i32, _p, C.c_uint64]),
    "lurkhip_cached_proof_bincode": (_i64, [_p, C.c_uint64, _p, _p, _p, C.c_uint64, _p, _i32, _p, C.c_uint64]),
    "lurkhip_proof_read": (_i32, [_p, _u32p, C.c_uint64]),
    "lurkhip_proof_free": (_i32, [_p]),
    "lurkhip_quotient_dev": (_i32, [_p, _p, C.c_uint32, _u32p, _u32p, _u32p, _u32p, _u32p, _u32p, _u32p, _u32p]),
    "lurkhip_comm_unique_id": (_i32, [_p]),
    "lurkhip_comm_create": (_i32, [_p, _p, _i32, _i32, C.POINTER(_p)]),
    "lurkhip_comm_destroy": (_i32, [_p, _p]),
    "lurkhip_comm_info": (_i32, [_p, C.POINTER(_i32), C.POINTER(_i32)]),
    "lurkhip_exchange_roots_dev": (_i32, [_p, _p, _u32p, _i32, _u32p]),
    "lurkhip_exchange_roots": (_i32, [_p, _p, _u32p, _u32p, _i32, _u32p]),
    "lurkhip_exchange_roots_var": (_i32, [_p, _p, _u32p, _u32p, _i32, _i32, _u32p]),
    "lurkhip_comm_library": (C.c_char_p, []),
    "lurkhip_reduce_sums_dev": (_i32, [_p, _p, _p, _u32p]),
    "lurkhip_reduce_sums": (_i32, [_p, _p, _u32p, _i32, _u32p]),
    "lurkhip_comm_split_vtable": (_i32, [_p, _p, _p]),
    "lurkhip_setup_split": (_i32, [_p, _p, _i32, _i32, _p, _u32p, _u32p, _i32, C.POINTER(_p), _u32p]),
    "lurkhip_shard_commit_split": (_i32, [_p, _p, _i32, _i32, _p, _u32p, _p, _u32p, _p, _i32, _i32, C.POINTER(_p), _u32p]),
    "lurkhip_func_trace_run_rows": (_i32, [_p, _p, C.c_uint32, C.c_uint32, _u32p, C.c_uint32, _i32]),
    "lurkhip_shard_prove_split": (_i32, [_p, _p, _p, _p, _u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_p)]),
    "lurkhip_prover_stats": (_i32, [_p, _p]),
    "lurkhip_split_stats": (_i32, [_p, _p, _i32]),
    "lurkhip_split_plan": (_i64, [_i32, _i32, _i32, _i32, _u32p, _u32p, _p, _u32p, _u32p, _u32p, _u32p, _u32p, _p, C.c_uint64]),
}


def _bind():
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args


_bind()


def last_error(ctx_handle) -> str:
    msg = lib.lurkhip_last_error(ctx_handle)
    return msg.decode("utf-8", "replace") if msg else ""


def check(status: int, ctx_handle=None) -> None:
    if status != OK:
        raise LurkHipError(status, last_error(ctx_handle))
