"""Host mirror of the reference device runtime (CudaRuntimeObj -> RocmRuntime).

Reference: include/cuda/cuda_runtime.h:60-140, src/cuda/cuda_runtime.cc; Python surface
`backend.CudaRuntime(device)`, `.init_comm(name, world, rank)` (src/ffi/ffi_infinitensor.cc:447-457).
"""
from __future__ import annotations

import ctypes as C
import enum

from ._lib import check, lib


class DType(enum.IntEnum):
    """reference: DataType::getIndex() (include/core/data_type.h:8-23)."""

    F32 = 1
    U8 = 2
    I8 = 3
    U16 = 4
    I16 = 5
    I32 = 6
    I64 = 7
    BOOL = 9
    F16 = 10
    F64 = 11
    U32 = 12
    U64 = 13
    BF16 = 16


class _DeviceInfo(C.Structure):
    _fields_ = [
        ("name", C.c_char * 64),
        ("arch", C.c_char * 32),
        ("compute_units", C.c_int),
        ("clock_mhz", C.c_int),
        ("memory_clock_mhz", C.c_int),
        ("memory_bus_bits", C.c_int),
        ("total_memory", C.c_size_t),
        ("wavefront_size", C.c_int),
        ("lds_bytes_per_cu", C.c_int),
    ]


class Event:
    def __init__(self):
        self._h = C.c_void_p()
        check(lib().infini_rocm_event_create(C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().infini_rocm_event_destroy(self._h)
            self._h = None


class Graph:
    """A captured, instantiated hipGraph (reference: CudaGraphCacheEntry, cuda_runtime.h:38-50)."""

    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        if getattr(self, "_h", None):
            lib().infini_rocm_graph_destroy(self._h)
            self._h = None


class RocmRuntime:
    """One device + one stream + one workspace (+ optionally one RCCL communicator)."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        check(lib().infini_rocm_runtime_create(int(device), C.byref(self._h)))
        self.device = int(device)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().infini_rocm_runtime_destroy(h)
            except Exception:
                pass
            self._h = None

    # -- identification -------------------------------------------------------------------
    @property
    def handle(self) -> C.c_void_p:
        return self._h

    def device_info(self) -> dict:
        info = _DeviceInfo()
        check(lib().infini_rocm_runtime_device_info(self._h, C.byref(info)))
        return {
            "name": info.name.decode(),
            "arch": info.arch.decode(),
            "compute_units": info.compute_units,
            "clock_mhz": info.clock_mhz,
            "memory_clock_mhz": info.memory_clock_mhz,
            "memory_bus_bits": info.memory_bus_bits,
            "total_memory": info.total_memory,
            "wavefront_size": info.wavefront_size,
            "lds_bytes_per_cu": info.lds_bytes_per_cu,
        }

    def to_string(self) -> str:  # reference: RuntimeObj::toString
        return "ROCM Runtime"

    # -- stream -----------------------------------------------------------------------------
    def stream(self) -> int:
        s = C.c_void_p()
        check(lib().infini_rocm_runtime_get_stream(self._h, C.byref(s)))
        return s.value or 0

    def set_stream(self, stream: int) -> None:
        """Adopt a native hipStream_t (0 = the legacy default stream)."""
        check(lib().infini_rocm_runtime_set_stream(self._h, C.c_void_p(stream or 0)))

    def use_own_stream(self) -> None:
        check(lib().infini_rocm_runtime_use_own_stream(self._h))

    def use_torch_stream(self) -> None:
        """Launch on torch's current stream so kernels order with torch allocations/copies."""
        import torch

        self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def wait_torch(self) -> None:
        """Order this runtime's stream behind torch's current stream (an event wait on the device, no host sync).

        A runtime created with a stream of its own does NOT order with torch: a tensor torch is still filling on its stream
        can be read too early by a kernel launched here (round 4: bench.py's Gather read an index tensor torch.randint had not
        written yet and faulted). Call this between building inputs with torch and launching, or use `use_torch_stream()`."""
        import torch

        s = self.stream()
        cur = torch.cuda.current_stream(self.device)
        if s == cur.cuda_stream:
            return
        if not s:  # the legacy default stream cannot be wrapped: a full wait
            cur.synchronize()
            return
        ev = torch.cuda.Event()
        ev.record(cur)
        torch.cuda.ExternalStream(s, device=self.device).wait_event(ev)

    def sync(self) -> None:
        check(lib().infini_rocm_runtime_sync(self._h))

    # -- memory -----------------------------------------------------------------------------
    def alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        check(lib().infini_rocm_alloc(self._h, nbytes, C.byref(p)))
        return p.value or 0

    def dealloc(self, ptr: int) -> None:
        check(lib().infini_rocm_dealloc(self._h, C.c_void_p(ptr)))

    def copy_from_cpu(self, dst: int, src, nbytes: int) -> None:
        check(lib().infini_rocm_copy_from_cpu(self._h, C.c_void_p(dst), src, nbytes))

    def copy_to_cpu(self, dst, src: int, nbytes: int) -> None:
        check(lib().infini_rocm_copy_to_cpu(self._h, dst, C.c_void_p(src), nbytes))

    def copy_inside(self, dst: int, src: int, nbytes: int) -> None:
        check(lib().infini_rocm_copy_inside(self._h, C.c_void_p(dst), C.c_void_p(src), nbytes))

    def workspace(self, nbytes: int) -> int:
        p = C.c_void_p()
        check(lib().infini_rocm_workspace(self._h, nbytes, C.byref(p)))
        return p.value or 0

    def workspace_info(self) -> dict:
        """Current scratch size, number of retired (outgrown, still allocated) blocks and the block epoch."""
        b, r, e = C.c_size_t(), C.c_size_t(), C.c_uint64()
        check(lib().infini_rocm_workspace_info(self._h, C.byref(b), C.byref(r), C.byref(e)))
        return {"bytes": b.value, "retired_blocks": r.value, "epoch": e.value}

    def workspace_trim(self) -> None:
        """Free retired scratch blocks (only when no captured graph of this runtime is alive)."""
        check(lib().infini_rocm_workspace_trim(self._h))

    # -- timing -----------------------------------------------------------------------------
    def record(self, ev: Event) -> None:
        check(lib().infini_rocm_event_record(self._h, ev._h))

    @staticmethod
    def elapsed_ms(start: Event, stop: Event) -> float:
        ms = C.c_float()
        check(lib().infini_rocm_event_elapsed_ms(start._h, stop._h, C.byref(ms)))
        return float(ms.value)

    # -- hipGraph ---------------------------------------------------------------------------
    def begin_capture(self) -> None:
        check(lib().infini_rocm_graph_begin_capture(self._h))

    def end_capture(self) -> Graph:
        g = C.c_void_p()
        check(lib().infini_rocm_graph_end_capture(self._h, C.byref(g)))
        return Graph(g)

    def abort_capture(self) -> None:
        check(lib().infini_rocm_graph_abort_capture(self._h))

    def launch_graph(self, graph: Graph) -> None:
        check(lib().infini_rocm_graph_launch(self._h, graph._h))

    # -- communicator (reference: CudaRuntimeObj::initComm, cuda_runtime.cc:495-509) ----------
    def init_comm(self, name: str, world_size: int, rank: int) -> None:
        check(lib().infini_rocm_comm_init(self._h, name.encode(), int(world_size), int(rank)))

    def init_comm_direct(self, name: str, world_size: int, rank: int) -> None:
        """The hand-written one-hop transport over IPC-mapped peer buffers (csrc/comm_direct.hip) — next to an RCCL
        communicator (pick with comm_set_algo) or alone; ranks may share a device."""
        check(lib().infini_rocm_comm_init_direct(self._h, name.encode(), int(world_size), int(rank)))

    def comm_set_algo(self, algo: int) -> None:
        """0: RCCL when initialised (default), 1: the direct transport."""
        check(lib().infini_rocm_comm_set_algo(self._h, int(algo)))

    def comm_check(self) -> None:
        """Raises when a direct-transport kernel gave up waiting for a peer (call after sync())."""
        check(lib().infini_rocm_comm_check(self._h))

    def init_comm_with_id(self, unique_id: bytes, world_size: int, rank: int) -> None:
        check(lib().infini_rocm_comm_init_id(self._h, unique_id, len(unique_id), int(world_size), int(rank)))

    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(256)
        n = C.c_size_t(256)
        check(lib().infini_rocm_comm_unique_id(buf, C.byref(n)))
        return buf.raw[: n.value]

    def comm_info(self) -> tuple[int, int]:
        w, r = C.c_int(), C.c_int()
        check(lib().infini_rocm_comm_info(self._h, C.byref(w), C.byref(r)))
        return w.value, r.value
