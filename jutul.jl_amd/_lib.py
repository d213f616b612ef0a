"""ctypes binding of libjutul_hip.so (the C ABI of include/jutul_hip.h).  No CPU fallback: if the shared
object or a GPU is missing, every compute entry point raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("JUTUL_HIP_LIB") or os.path.join(_HERE, "libjutul_hip.so")  # (override: A/B of two builds in one GPU session)
_lib = None

I64P = C.POINTER(C.c_int64)
I32P = C.POINTER(C.c_int32)
F64P = C.POINTER(C.c_double)
H = C.c_void_p


class JutulHIPError(RuntimeError):
    pass


class NewtonReport(C.Structure):
    _fields_ = [("assembly_ms", C.c_double), ("convergence_ms", C.c_double), ("precond_ms", C.c_double),
                ("linear_solve_ms", C.c_double), ("update_ms", C.c_double), ("error", C.c_double * 2),
                ("lin_res0", C.c_double), ("lin_res", C.c_double), ("linear_iterations", C.c_int64),
                ("linear_status", C.c_int32), ("converged", C.c_int32)]


# name -> argtypes (restype is always int32).  Mirrors include/jutul_hip.h one to one.
SIGNATURES = {
    "jh_last_error": [C.c_char_p, C.c_int64],
    "jh_version": [],
    "jh_context_create": [C.c_int32, C.POINTER(H)],
    "jh_context_create_host": [C.POINTER(H)],
    "jh_context_plan_checksum": [H, C.POINTER(C.c_int64)],
    "jh_context_destroy": [H],
    "jh_synchronize": [H],
    "jh_context_set_option": [H, C.c_char_p, C.c_int64],
    "jh_context_get_option": [H, C.c_char_p, C.POINTER(C.c_int64)],
    "jh_context_set_cu_mask": [H, C.c_int32, C.c_int32],
    "jh_timer_start": [H],
    "jh_timer_stop_ms": [H, F64P],
    "jh_tpfa_create": [H, C.c_int64, C.c_int64, I64P, C.c_int32, C.c_int32, I64P, C.c_int64, C.c_int64, C.POINTER(H)],
    "jh_tpfa_create_weighted": [H, C.c_int64, C.c_int64, I64P, F64P, C.c_int32, C.c_int32, I64P, C.c_int64, C.c_int64, C.POINTER(H)],
    "jh_tpfa_destroy": [H],
    "jh_tpfa_sizes": [H, I64P, I64P, I64P, I64P, I32P],
    "jh_tpfa_get_conn": [H, I64P, I64P, I64P, I64P, I64P],
    "jh_tpfa_get_pattern": [H, I64P, I64P],
    "jh_tpfa_get_positions": [H, I64P, I64P],
    "jh_tpfa_get_pattern_layout": [H, C.c_int32, I64P, I64P],
    "jh_tpfa_get_positions_layout": [H, C.c_int32, I64P, I64P],
    "jh_tpfa_get_ordering": [H, I64P, I64P, I64P, C.c_int64],
    "jh_tpfa_get_split": [H, I64P, I64P, I64P],
    "jh_krylov_set_min_iterations": [H, C.c_int64],
    "jh_vec_create": [H, C.POINTER(H)],
    "jh_vec_create_for": [H, C.POINTER(H)],
    "jh_vec_destroy": [H],
    "jh_vec_upload": [H, F64P],
    "jh_vec_download": [H, F64P],
    "jh_vec_fill": [H, C.c_double],
    "jh_vec_copy": [H, H],
    "jh_vec_axpby": [H, C.c_double, H, C.c_double],
    "jh_vec_dot": [H, H, F64P],
    "jh_vec_length": [H, I64P],
    "jh_vec_negate_into": [H, H],
    "jh_csr_create": [H, C.POINTER(H)],
    "jh_csr_create_from_pattern": [H, C.c_int64, C.c_int32, I64P, I64P, F64P, C.POINTER(H)],
    "jh_csr_destroy": [H],
    "jh_csr_sizes": [H, I64P, I64P, I32P],
    "jh_csr_set_values": [H, F64P],
    "jh_csr_get_values": [H, F64P],
    "jh_csr_get_values_layout": [H, C.c_int32, F64P],
    "jh_csr_set_values_layout": [H, C.c_int32, F64P],
    "jh_vec_upload_layout": [H, C.c_int32, F64P],
    "jh_vec_download_layout": [H, C.c_int32, F64P],
    "jh_spmv": [H, H, H, C.c_double, C.c_double],
    "jh_spmv_jagged": [H, H, H, C.c_double, C.c_double],
    "jh_spmv_info": [H, I64P],
    "jh_scale_system": [H, H, C.c_int32, C.c_double],
    "jh_unit_diagonalize": [H, H, C.c_int64],
    "jh_law_create": [H, C.c_int32, F64P, C.POINTER(H)],
    "jh_law_create_custom": [H, C.c_char_p, F64P, C.c_int32, C.POINTER(H)],
    "jh_law_destroy": [H],
    "jh_law_set_data": [H, C.c_int32, F64P],
    "jh_law_set_state": [H, F64P],
    "jh_law_set_state0": [H, F64P],
    "jh_law_get_state": [H, F64P],
    "jh_law_get_variable": [H, C.c_int32, C.c_int32, F64P],
    "jh_host_register": [C.c_void_p, C.c_int64],
    "jh_host_unregister": [C.c_void_p],
    "jh_law_update_state0": [H],
    "jh_law_reset_state": [H],
    "jh_law_set_sources": [H, C.c_int64, I64P, F64P],
    "jh_assemble": [H, C.c_double, H, H],
    "jh_convergence": [H, H, C.c_int64, F64P],
    "jh_law_set_update_limits": [H, F64P],
    "jh_update_primary": [H, H, C.c_double, F64P],
    "jh_increment_norm": [H, H, C.c_int64, F64P],
    "jh_law_change_report": [H, C.c_int64, F64P],
    "jh_ilu0_create": [H, I64P, C.c_int64, C.POINTER(H)],
    "jh_ilu0_destroy": [H],
    "jh_ilu0_factor": [H],
    "jh_ilu0_apply": [H, H, H],
    "jh_ilu0_apply_mul": [H, H, H, H],
    "jh_ilu0_get_factor": [H, F64P],
    "jh_ilu0_info": [H, I64P, I64P, I64P],
    "jh_ilu0_stats": [H, I64P],
    "jh_diag_precond_create": [H, C.c_int32, C.c_double, C.POINTER(H)],
    "jh_krylov_create": [H, C.POINTER(H)],
    "jh_krylov_destroy": [H],
    "jh_krylov_last_path": [H, I64P],
    "jh_krylov_profile": [H, C.c_int32, C.c_int32, F64P, I64P],
    "jh_bicgstab": [H, H, C.c_int32, H, H, C.c_double, C.c_double, C.c_int64, I64P, I32P, F64P, C.c_int64],
    "jh_gmres": [H, H, C.c_int32, H, H, C.c_double, C.c_double, C.c_int64, I64P, I32P, F64P, C.c_int64],
    "jh_newton_step": [H, H, H, H, H, H, C.c_double, C.c_double, C.c_int32, C.c_double, C.c_double, C.c_int64,
                       C.c_int32, C.POINTER(NewtonReport)],
    "jh_comm_unique_id": [C.c_char_p],
    "jh_comm_init": [H, C.c_int32, C.c_int32, C.c_char_p],
    "jh_comm_finalize": [H],
    "jh_comm_init_ipc_only": [H, C.c_int32, C.c_int32],
    "jh_comm_set_halo_callback": [H, C.c_void_p, C.c_void_p],
    "jh_comm_info": [H, I64P],
    "jh_halo_info": [H, I64P],
    "jh_comm_ipc_export": [H, C.c_char_p],
    "jh_comm_ipc_attach": [H, C.c_char_p, C.POINTER(C.c_int32)],
    "jh_comm_ipc_enable": [H, C.c_int32],
    "jh_comm_set_exclusive": [H, C.c_int32],
    "jh_comm_devices_distinct": [H, C.POINTER(C.c_int32)],
    "jh_comm_xrank_selftest": [H, C.POINTER(C.c_int32)],
    "jh_halo_ipc_export": [H, C.c_char_p],
    "jh_halo_ipc_attach": [H, C.c_char_p, I64P, I64P, C.POINTER(C.c_int32)],
    "jh_halo_ipc_selftest": [H, H, F64P, C.POINTER(C.c_int32)],
    "jh_halo_ipc_enable": [H, C.c_int32],
    "jh_comm_local_group_create": [C.c_int32, C.POINTER(H)],
    "jh_comm_local_group_destroy": [H],
    "jh_comm_init_local": [H, H, C.c_int32],
    "jh_halo_create": [H, C.c_int64, C.c_int32, I32P, I64P, I64P, I64P, I64P],
    "jh_halo_exchange": [H, H],
    "jh_halo_exchange_state": [H],
    "jh_allreduce": [H, F64P, C.c_int32, C.c_int32],
    "jh_partition_graph": [C.c_int64, C.c_int64, I64P, F64P, C.c_int64, C.c_double, I64P],
    "jh_partition_rcb": [C.c_int64, C.c_int32, F64P, C.c_int64, I64P],
    "jh_subdomain_create": [C.c_int64, C.c_int64, I64P, I64P, C.c_int64, C.c_int32, C.POINTER(H)],
    "jh_subdomain_sizes": [H, I64P],
    "jh_subdomain_get": [H, I64P, I64P, I64P, C.POINTER(C.c_int32), I64P, I64P, I64P, I64P],
    "jh_subdomain_destroy": [H],
}


def load():
    """Loads libjutul_hip.so (import torch first so that the process uses a single HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        try:  # the build is in-tree and cheap (hipcc cross-compiles gfx950 without a GPU)
            import importlib.util
            spec = importlib.util.spec_from_file_location("jh_build", os.path.join(_HERE, "build.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            mod.build()
        except Exception as e:  # noqa: BLE001
            raise JutulHIPError(f"{SO_PATH} is missing and could not be built ({e}); run "
                                "`python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)") from e
    try:
        import torch  # noqa: F401  (loads libamdhip64 / librccl that torch ships; ours then binds to the same)
    except Exception:
        pass
    lib = C.CDLL(SO_PATH, mode=C.RTLD_GLOBAL)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int32
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        buf = C.create_string_buffer(4096)
        load().jh_last_error(buf, 4096)
        raise JutulHIPError(buf.value.decode(errors="replace"))


def i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def pi(a):
    return a.ctypes.data_as(I64P) if a is not None else None


def pi32(a):
    return a.ctypes.data_as(I32P) if a is not None else None


def pf(a):
    return a.ctypes.data_as(F64P) if a is not None else None
