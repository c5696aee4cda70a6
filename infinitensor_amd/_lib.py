"""ctypes binding of the C ABI declared in include/infini_rocm.h.

The HIP library is the product: if it is missing or fails to load this module raises — there is
no CPU or PyTorch fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from pathlib import Path

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / "lib" / "libinfini_rocm.so"
HEADER = PKG.parent / "include" / "infini_rocm.h"

_lib = None


class InfiniRocmError(RuntimeError):
    """Raised for every non-zero status (reference: infini::Exception -> Python RuntimeError)."""


def declared_symbols() -> list[str]:
    """Every function name declared in include/infini_rocm.h."""
    text = HEADER.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(infini_rocm_\w+)\s*\(", text)))


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("INFINI_ROCM_LIB", LIB_PATH))
    if not path.exists():
        raise ImportError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no fallback path."
        )
    # torch ships its own libamdhip64.so.7 / librccl.so.1 (same SONAMEs as /opt/rocm). Import it
    # FIRST so the dynamic loader binds our library to the HIP runtime torch already loaded;
    # the other order maps two HIP runtimes into one process (torch asks for the un-versioned
    # name, which does not match the loaded SONAME).
    import torch  # noqa: F401  (device-memory plumbing; never used for compute)

    L = C.CDLL(str(path), mode=C.RTLD_GLOBAL)
    vp, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
    pvp = C.POINTER(C.c_void_p)
    pi64 = C.POINTER(C.c_int64)

    def sig(name, args, res=C.c_int):
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = res

    sig("infini_rocm_last_error", [], C.c_char_p)
    sig("infini_rocm_version", [], C.c_char_p)
    sig("infini_rocm_device_count", [C.POINTER(i32)])
    sig("infini_rocm_runtime_create", [i32, pvp])
    sig("infini_rocm_runtime_destroy", [vp])
    sig("infini_rocm_runtime_device_info", [vp, vp])
    sig("infini_rocm_runtime_get_stream", [vp, pvp])
    sig("infini_rocm_runtime_set_stream", [vp, vp])
    sig("infini_rocm_runtime_use_own_stream", [vp])
    sig("infini_rocm_runtime_sync", [vp])
    sig("infini_rocm_alloc", [vp, sz, pvp])
    sig("infini_rocm_dealloc", [vp, vp])
    sig("infini_rocm_copy_from_cpu", [vp, vp, vp, sz])
    sig("infini_rocm_copy_to_cpu", [vp, vp, vp, sz])
    sig("infini_rocm_copy_inside", [vp, vp, vp, sz])
    sig("infini_rocm_memset", [vp, vp, i32, sz])
    sig("infini_rocm_workspace", [vp, sz, pvp])
    sig("infini_rocm_workspace_trim", [vp])
    sig("infini_rocm_workspace_info", [vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(C.c_uint64)])
    sig("infini_rocm_probe_mfma_ceiling", [vp, i32, vp, vp, i32, C.POINTER(C.c_double)])
    sig("infini_rocm_probe_mfma_ceiling32", [vp, i32, vp, vp, i32, C.POINTER(C.c_double)])
    sig("infini_rocm_probe_mfma_a_from_l2", [vp, i32, vp, vp, vp, i32, i32, C.POINTER(C.c_double)])
    sig("infini_rocm_probe_mfma_wave128", [vp, i32, vp, vp, i32, i32, C.POINTER(C.c_double)])
    sig("infini_rocm_probe_gemm_timeline", [vp, vp, vp, vp, i64, i64, i64, i32, vp])
    sig("infini_rocm_event_create", [pvp])
    sig("infini_rocm_event_destroy", [vp])
    sig("infini_rocm_event_record", [vp, vp])
    sig("infini_rocm_event_elapsed_ms", [vp, vp, C.POINTER(f32)])
    sig("infini_rocm_graph_begin_capture", [vp])
    sig("infini_rocm_graph_end_capture", [vp, pvp])
    sig("infini_rocm_graph_abort_capture", [vp])
    sig("infini_rocm_graph_launch", [vp, vp])
    sig("infini_rocm_graph_destroy", [vp])
    sig("infini_rocm_matmul", [vp, i32, vp, vp, vp, vp, i64, i64, i64, i64, i32, i32, i64, i64, i64, i64, i64, i32])
    sig("infini_rocm_matmul_headsplit", [vp, i32, vp, vp, vp, vp, i64, i64, i64, i64, i32, i32, i64, i64, i64, i64, i64, i32, i64, i64])
    sig("infini_rocm_matmul_grouped", [vp, i32, vp, vp, vp, vp, i64, i64, i64, i64, i32, i32, i64, i64, i64, i64, i64, i64, i32, i64, i64])
    sig("infini_rocm_matmul_set_variant", [vp, i32])
    sig("infini_rocm_matmul_last_variant", [vp, C.POINTER(C.c_int)])
    sig("infini_rocm_matmul_may_use_workspace", [vp, i64, i64, i64, C.POINTER(C.c_int)])
    sig("infini_rocm_matmul_num_variants", [], i32)
    sig("infini_rocm_matmul_variant_name", [i32], C.c_char_p)
    sig("infini_rocm_softmax", [vp, i32, vp, vp, i64, i64, i64])
    sig("infini_rocm_layer_norm", [vp, i32, vp, vp, vp, vp, i64, i64, i64, i64, f32])
    sig("infini_rocm_rms_norm", [vp, i32, vp, vp, vp, i64, i64, f32])
    sig("infini_rocm_conv2d_set_variant", [vp, i32])
    sig("infini_rocm_conv2d_last_route", [vp, C.POINTER(C.c_char_p)])
    sig("infini_rocm_conv2d_pool_supported", [i32, i64, i64, i64, i64, i64, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32])
    sig("infini_rocm_conv2d_pool", [vp, i32, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32])
    sig("infini_rocm_conv2d_set_const_weights", [vp, i32])
    sig("infini_rocm_weight_cache_info", [vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(C.c_uint64)])
    sig("infini_rocm_weight_cache_clear", [vp])
    sig("infini_rocm_attention", [vp, i32, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, vp, i32, f32, i32])
    sig("infini_rocm_attention_ex", [vp, i32, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, vp, i32, f32, i32, i64, i32])
    sig("infini_rocm_attention_kvcache", [vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, i64, i64, i64])
    sig("infini_rocm_gather_elements", [vp, i32, i32, vp, vp, vp, i32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), i32])
    sig("infini_rocm_resize", [vp, i32, vp, vp, i32, pi64, pi64, C.POINTER(C.c_float), C.POINTER(C.c_float), i32, i32, i32])
    sig("infini_rocm_conv_transpose2d", [vp, i32, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i32, i32, i32, i32, i32, i32, i32, i32, i64, i32])
    sig("infini_rocm_conv2d_res", [vp, i32, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i32, i32, i32, i32, i32, i32, i64, i32])
    sig("infini_rocm_bias_residual", [vp, i32, vp, vp, vp, vp, i64, i64, i64, i32])
    sig("infini_rocm_add_norm", [vp, i32, i32, vp, vp, vp, vp, vp, i64, i64, i64, i64, f32])
    sig("infini_rocm_bias_add_norm", [vp, i32, i32, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, f32])
    sig("infini_rocm_pool2d_relu", [vp, i32, i32, vp, vp, i64, i64, i64, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32])
    sig("infini_rocm_rope", [vp, i32, i32, vp, vp, vp, i64, i64, i64, f32])
    sig("infini_rocm_rope_headsplit", [vp, i32, i32, vp, vp, vp, i64, i64, i64, f32, i64])
    sig("infini_rocm_binary", [vp, i32, i32, vp, vp, vp, i32, pi64, pi64, pi64])
    sig("infini_rocm_unary", [vp, i32, i32, vp, vp, i64, f32, f32])
    sig("infini_rocm_silu_mul", [vp, i32, vp, vp, vp, i64])
    sig("infini_rocm_cast", [vp, i32, i32, vp, vp, i64])
    sig("infini_rocm_conv2d", [vp, i32, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i32, i32, i32, i32, i32, i32, i64, i32])
    sig("infini_rocm_reduce", [vp, i32, i32, vp, vp, i32, pi64, C.POINTER(i32)])
    sig("infini_rocm_lrn", [vp, i32, vp, vp, i64, i64, i64, i32, f32, f32, f32])
    sig("infini_rocm_batch_norm", [vp, i32, vp, vp, vp, vp, vp, vp, i64, i64, i64, f32])
    sig("infini_rocm_pool2d", [vp, i32, i32, vp, vp, i64, i64, i64, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32])
    sig("infini_rocm_transpose", [vp, i32, vp, vp, i32, pi64, C.POINTER(i32)])
    sig("infini_rocm_expand", [vp, i32, vp, vp, i32, pi64, pi64])
    sig("infini_rocm_gather", [vp, i32, i32, vp, vp, vp, i64, i64, i64, i64])
    sig("infini_rocm_where", [vp, i32, vp, vp, vp, vp, i32, pi64, pi64, pi64, pi64])
    sig("infini_rocm_where_ex", [vp, i32, i32, vp, vp, vp, vp, i32, pi64, pi64, pi64, pi64])
    sig("infini_rocm_pad_slice", [vp, i32, vp, vp, i32, pi64, pi64, pi64, pi64, i32])
    sig("infini_rocm_strided_copy", [vp, vp, vp, i64, i64, i64, i64])
    sig("infini_rocm_strided_copy_multi", [vp, i32, vp, vp, i64, vp, vp, vp])
    # "from-shape" forms (csrc/shaped.hip): the shape -> stride glue below the ABI, shared with the plugin kernels
    sig("infini_rocm_broadcast_strides", [i32, pi64, i32, pi64, pi64])
    sig("infini_rocm_binary_shaped", [vp, i32, i32, vp, i32, pi64, vp, i32, pi64, vp, i32, pi64])
    sig("infini_rocm_where_shaped", [vp, i32, i32, vp, i32, pi64, vp, i32, pi64, vp, i32, pi64, vp, i32, pi64])
    sig("infini_rocm_expand_shaped", [vp, i32, vp, i32, pi64, vp, i32, pi64])
    sig("infini_rocm_matmul_plan", [i32, pi64, i32, pi64, i32, pi64, i32, i32, pi64])
    sig("infini_rocm_matmul_shaped", [vp, i32, vp, i32, pi64, vp, i32, pi64, vp, i32, pi64, vp, i32, i32, i32, i64, i64])
    sig("infini_rocm_concat_shaped", [vp, i32, i32, vp, pi64, vp, i32, pi64, i32])
    sig("infini_rocm_split_shaped", [vp, i32, i32, vp, pi64, vp, i32, pi64, i32])
    sig("infini_rocm_pad_shaped", [vp, i32, vp, vp, i32, pi64, pi64])
    sig("infini_rocm_gather_shaped", [vp, i32, i32, vp, i32, pi64, vp, i64, vp, i32])
    sig("infini_rocm_comm_init", [vp, C.c_char_p, i32, i32])
    sig("infini_rocm_comm_unique_id", [vp, C.POINTER(sz)])
    sig("infini_rocm_comm_init_id", [vp, vp, sz, i32, i32])
    sig("infini_rocm_matmul_set_compute_type", [vp, i32])
    sig("infini_rocm_comm_init_direct", [vp, C.c_char_p, i32, i32])
    sig("infini_rocm_comm_set_algo", [vp, i32])
    sig("infini_rocm_comm_check", [vp])
    sig("infini_rocm_comm_destroy", [vp])
    sig("infini_rocm_comm_info", [vp, C.POINTER(i32), C.POINTER(i32)])
    sig("infini_rocm_all_reduce", [vp, i32, i32, vp, vp, i64])
    sig("infini_rocm_all_reduce_async", [vp, i32, i32, vp, vp, i64])
    sig("infini_rocm_comm_join", [vp])
    sig("infini_rocm_reduce_scatter", [vp, i32, vp, vp, i64, i32])
    sig("infini_rocm_all_gather", [vp, i32, vp, vp, i64])
    sig("infini_rocm_broadcast", [vp, i32, vp, vp, i64, i32])
    sig("infini_rocm_send", [vp, i32, vp, i64, i32])
    sig("infini_rocm_recv", [vp, i32, vp, i64, i32])
    _lib = L
    return L


def check(status: int) -> None:
    if status != 0:
        msg = lib().infini_rocm_last_error().decode(errors="replace")
        raise InfiniRocmError(f"infini_rocm status {status}: {msg}")
