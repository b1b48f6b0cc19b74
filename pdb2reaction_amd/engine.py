"""ctypes binding of ``libumx.so`` (C ABI in ``include/umx.h``).

The library is the product: if it is missing or no gfx950 device is usable this module raises --
there is no CPU fallback (the CPU restatement under ``oracle/`` is test infrastructure only).
"""
from __future__ import annotations

import ctypes as C
import os
import warnings
from typing import Dict, Optional, Sequence, Tuple, Union

import numpy as np

from . import weights as W

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libumx.so")

_lib = None


class ProfileStats(C.Structure):
    """Mirror of ``umx_profile_stats`` (include/umx.h): [0] split-precision plane GEMM family, [1] fp32-MFMA GEMM family, [2] fused
    radial-MLP kernels."""

    _fields_ = [("ms", C.c_double * 3), ("launches", C.c_int64 * 3), ("alg_flops", C.c_double * 3), ("mfma_flops", C.c_double * 3)]


UMX_ERR_RANGE = -6      # include/umx.h


class UmxError(RuntimeError):
    """A libumx call returned a non-zero status."""


def _init_torch_hip_first() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  If libumx.so (linked against /opt/rocm) brings
    the GPU up first, a later ``torch.cuda`` initialisation in the same process finds "No HIP GPUs are available" (two HSA
    runtimes, the second cannot open the device).  Initialising torch's runtime first makes both share one; without torch
    (plain C consumers, CPU-only torch) there is nothing to order."""
    try:
        import torch

        if getattr(torch.version, "hip", None) and torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen libumx.so and declare every entry point of include/umx.h."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("UMX_LIBRARY", LIB_PATH)
    if not os.path.exists(p):
        raise ImportError(
            f"{p} not found: the HIP engine is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `python -m pdb2reaction_amd.build`) in the repository root; there is no CPU fallback."
        )
    _init_torch_hip_first()
    lib = C.CDLL(p)
    vp, i32, i64p, dp, fp = C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_float)
    sigs = {
        "umx_abi_version": ([], i32),
        "umx_build_digest": ([], C.c_char_p),
        "umx_create": ([C.POINTER(vp), i32], i32),
        "umx_destroy": ([vp], i32),
        "umx_last_error": ([vp], C.c_char_p),
        "umx_load_weights": ([vp, vp, C.c_size_t], i32),
        "umx_set_precision": ([vp, C.c_char_p], i32),
        "umx_precision_mode": ([vp], C.c_char_p),
        "umx_model_variant": ([vp], C.c_char_p),
        "umx_set_system": ([vp, i32, C.POINTER(C.c_int32), i32, i32, i32, C.c_float, i32], i32),
        "umx_set_workspace_limit": ([vp, C.c_size_t], i32),
        "umx_energy_forces": ([vp, i32, fp, dp, fp], i32),
        "umx_energy_forces_dev": ([vp, i32, vp, vp, vp, vp], i32),
        "umx_gp_begin": ([vp, vp, i32, i32, vp, vp, vp], i32),
        "umx_gp_step": ([vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(i32)], i32),
        "umx_synchronize": ([vp], i32),
        "umx_last_graph_stats": ([vp, i64p, C.POINTER(C.c_int32)], i32),
        "umx_last_partitions": ([vp], i32),
        "umx_last_lanes": ([vp], i32),
        "umx_reserve_images": ([vp, i32], i32),
        "umx_workspace_stats": ([vp, C.POINTER(C.c_int64), C.POINTER(C.c_int32)], i32),
        "umx_profile_enable": ([vp, i32], i32),
        "umx_profile_read": ([vp, C.POINTER(ProfileStats), i32], i32),
        "umx_bond_changes": ([vp, i32, dp, dp, dp, C.c_double, C.c_double, C.c_double, dp, dp, C.POINTER(C.c_uint8)], i32),
        "umx_debug_fetch": ([vp, C.c_char_p, vp, C.c_size_t, C.POINTER(C.c_size_t)], i32),
        "umx_debug_keep": ([vp, i32], i32),
    }
    for name, (args, res) in sigs.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.argtypes, fn.restype = args, res
    if path is None:
        # content check against the sources in this tree: the .so is git-ignored and travels prebuilt, so an edit without a
        # rebuild would silently run old kernels (VERDICT r1 "stale-.so hazard")
        from .build import dependencies, source_digest

        have = lib.umx_build_digest().decode()
        if all(os.path.exists(d) for d in dependencies()) and any(d.endswith(".hip") for d in dependencies()):
            want, what = source_digest(), "tree"
        else:
            # a deployment that ships the prebuilt library without csrc/ and include/ (ADVICE r2): nothing to hash -- hold the
            # library to the digest file written next to it at build time, or accept it with a warning when that is absent too
            try:
                with open(p + ".digest") as f:
                    want, what = f.read().strip(), "libumx.so.digest"
            except OSError:
                want, what = have, "none"
                warnings.warn(f"{p}: neither the kernel sources nor libumx.so.digest are present; the library's build digest "
                              f"({have[:12]}...) cannot be checked", RuntimeWarning)
        if have != want and os.environ.get("UMX_ALLOW_STALE", "0") != "1":
            raise ImportError(f"{p} was built from other sources (digest {have[:12]}..., {what} {want[:12]}...): rebuild with "
                              "`python -m pdb2reaction_amd.build` (UMX_ALLOW_STALE=1 overrides)")
        _lib = lib
    return lib


EXPORTED_SYMBOLS = (
    "umx_abi_version", "umx_build_digest", "umx_create", "umx_destroy", "umx_last_error", "umx_load_weights", "umx_set_precision", "umx_precision_mode", "umx_model_variant", "umx_set_system",
    "umx_set_workspace_limit", "umx_energy_forces", "umx_energy_forces_dev", "umx_gp_begin", "umx_gp_step", "umx_synchronize",
    "umx_last_graph_stats", "umx_last_partitions", "umx_last_lanes", "umx_reserve_images", "umx_workspace_stats", "umx_profile_enable", "umx_profile_read", "umx_bond_changes", "umx_debug_fetch", "umx_debug_keep",
)


class Engine:
    """One UMA-S engine on one GPU (one per process/rank)."""

    def __init__(self, device: int = 0, precision: Optional[str] = None):
        """precision: None = the UMX_PRECISION environment variable (default "auto" = "bf16x3": 3 x 3 bf16 planes / 6 products in both
        passes, the like-for-like arithmetic to float32 -- include/umx.h), else "auto" | "bf16x3" | "split" (the opt-in fast mode:
        fp16 forward planes, 16-bit reverse) | "split-bf16" | "fp32"."""
        self.lib = load_library()
        self._h = C.c_void_p()
        st = self.lib.umx_create(C.byref(self._h), int(device))
        if st != 0:
            raise UmxError(f"umx_create failed ({st}): {self.lib.umx_last_error(None).decode()}")
        self.device = int(device)
        self.natoms = 0
        self.precision = precision
        self.widened = False            # True once an fp16 range violation moved this engine to bf16 forward planes (split-bf16 / bf16x3)
        self._blob = None
        self._system = None
        self.dataset_list = tuple(W.DATASET_LIST)      # order of the rows of the loaded blob's dataset_embedding.weight (load_weights)
        if precision is not None:
            self._chk(self.lib.umx_set_precision(self._h, precision.encode()), "umx_set_precision")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.umx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st: int, what: str):
        if st != 0:
            err = UmxError(f"{what} failed ({st}): {self.lib.umx_last_error(self._h).decode()}")
            err.status = st
            raise err

    # ---- setup -----------------------------------------------------------------------------------
    def load_weights(self, weights: Union[bytes, Dict[str, np.ndarray]]):
        blob = weights if isinstance(weights, (bytes, bytearray)) else W.pack_blob(weights)
        self._blob = bytes(blob)        # kept (~27 MB) so that a range violation can re-load the engine in split-bf16
        buf = C.create_string_buffer(self._blob, len(self._blob))
        self._chk(self.lib.umx_load_weights(self._h, C.cast(buf, C.c_void_p), len(self._blob)), "umx_load_weights")
        # task names -> rows of dataset_embedding.weight: the checkpoint's own dataset_list when the blob records one (checkpoint.py), else UMA's
        meta = weights.meta if hasattr(weights, "meta") else W.blob_meta(self._blob)
        self.dataset_list = tuple((meta.get("model") or {}).get("dataset_list") or W.DATASET_LIST)
        # the list must name exactly the rows the blob's dataset embedding has (a re-ordered or shorter table without a trailer would
        # otherwise map task names to the wrong rows silently); datasets=0: a model without dataset embedding takes any task name
        try:
            n_rows = int(self.model_variant().rsplit("datasets=", 1)[1].split(";")[0])
        except (IndexError, ValueError):
            n_rows = len(self.dataset_list)
        if n_rows and n_rows != len(self.dataset_list):
            raise UmxError(f"the weight blob's dataset embedding has {n_rows} rows but its dataset_list names {len(self.dataset_list)} tasks "
                           f"{self.dataset_list}: task names cannot be mapped to rows (convert the checkpoint with its dataset_list)")

    def model_variant(self) -> str:
        """"ff=spectral|grid(G=..);emb=rand_emb|pos_emb|lin_emb;datasets=N" -- the model variant the loaded blob is (``umx_model_variant``)."""
        return self.lib.umx_model_variant(self._h).decode()

    def set_system(self, atomic_numbers: Sequence[int], charge: int = 0, spin: int = 1, task: str = "omol",
                   radius: Optional[float] = None, max_neigh: Optional[int] = None):
        z = np.ascontiguousarray(atomic_numbers, dtype=np.int32)
        if task not in self.dataset_list:
            raise ValueError(f"task_name {task!r} not in {self.dataset_list}")
        self._chk(self.lib.umx_set_system(self._h, len(z), z.ctypes.data_as(C.POINTER(C.c_int32)), int(charge), int(spin),
                                          self.dataset_list.index(task), float(radius or 0.0), int(max_neigh or 0)),
                  "umx_set_system")
        self.natoms = len(z)
        self._system = (z.copy(), int(charge), int(spin), task, radius, max_neigh)

    def set_workspace_limit(self, nbytes: int):
        self._chk(self.lib.umx_set_workspace_limit(self._h, int(nbytes)), "umx_set_workspace_limit")

    # ---- evaluation ------------------------------------------------------------------------------
    def energy_forces(self, pos_ang: np.ndarray, forces: bool = True) -> Tuple[np.ndarray, Optional[np.ndarray]]:
        """pos_ang: (K,N,3) or (N,3) Angstrom -> (E [K] eV float64, F [K,N,3] eV/A float32 | None)."""
        p = np.ascontiguousarray(pos_ang, dtype=np.float32)
        if p.ndim == 2:
            p = p[None]
        if p.ndim != 3 or p.shape[1] != self.natoms or p.shape[2] != 3:
            raise ValueError(f"positions must be (K,{self.natoms},3), got {p.shape}")
        k = p.shape[0]
        e = np.empty(k, dtype=np.float64)
        f = np.empty_like(p) if forces else None
        fp = C.POINTER(C.c_float)
        try:
            self._chk(self.lib.umx_energy_forces(self._h, k, p.ctypes.data_as(fp), e.ctypes.data_as(C.POINTER(C.c_double)),
                                                 f.ctypes.data_as(fp) if forces else None), "umx_energy_forces")
        except UmxError as err:
            # UMX_ERR_RANGE in the default mode (the input was finite, the entry checks): an activation left the fp16 operand range.  Same HIP path,
            # wider operands: re-load this engine with three bf16 forward planes (float32's range) and evaluate again.
            if getattr(err, "status", 0) != UMX_ERR_RANGE or not self._widen(str(err)):
                raise
            self._chk(self.lib.umx_energy_forces(self._h, k, p.ctypes.data_as(fp), e.ctypes.data_as(C.POINTER(C.c_double)),
                                                 f.ctypes.data_as(fp) if forces else None), "umx_energy_forces")
        return e, f

    def precision_mode(self) -> str:
        """The arithmetic the engine is in now ("bf16x3" | "split-f16" | "split-bf16" | "fp32"): what "auto" resolved to."""
        return self.lib.umx_precision_mode(self._h).decode()

    def take_range_error(self) -> bool:
        """Synchronise and collect the sticky range flag of the device-pointer entries: True when an evaluation since the last
        check produced a non-finite energy (the flag is cleared), False otherwise; any other failure raises."""
        st = self.lib.umx_synchronize(self._h)
        if st == UMX_ERR_RANGE:
            return True
        self._chk(st, "umx_synchronize")
        return False

    def widen(self, why: str = "range violation reported by a peer rank") -> bool:
        """Public form of the fp16 -> bf16 forward-plane switch, for callers that decide it collectively (parallel.py, hessian.py)."""
        return self._widen(why)

    _WIDER = {"split-f16": "split-bf16"}       # the mode with the same reverse pass and bf16 (float32-range) forward planes

    def _widen(self, why: str) -> bool:
        """Move an engine whose forward operands are fp16 planes (split-f16) to the mode with bf16 forward planes and the same reverse
        pass (split-bf16) -- once; False when the engine is not in such a mode (or UMX_NO_WIDEN=1)."""
        wider = self._WIDER.get(self.precision_mode())
        if self.widened or self._blob is None or self._system is None or wider is None:
            return False
        if os.environ.get("UMX_NO_WIDEN", "0") == "1":
            return False
        warnings.warn(f"pdb2reaction_amd: {why} -- re-loading the engine with bf16 forward planes (UMX_PRECISION={wider})", RuntimeWarning)
        self._chk(self.lib.umx_set_precision(self._h, wider.encode()), "umx_set_precision")
        self.precision, self.widened = wider, True
        self.load_weights(self._blob)
        z, charge, spin, task, radius, max_neigh = self._system
        self.set_system(z, charge, spin, task, radius, max_neigh)
        return True

    def energy_forces_dev(self, n_images: int, d_pos: int, d_energy: int, d_forces: Optional[int], stream: int = 0):
        """Device-pointer form (integers from e.g. ``tensor.data_ptr()``); enqueues on ``stream``, a ``hipStream_t`` handle
        as returned by ``torch.cuda.current_stream().cuda_stream``.  0 is the legacy default stream (torch's default
        stream), NOT a private engine stream: consumers on the same stream need no further synchronisation."""
        self._chk(self.lib.umx_energy_forces_dev(self._h, int(n_images), C.c_void_p(d_pos), C.c_void_p(d_energy),
                                                 C.c_void_p(d_forces) if d_forces else None,
                                                 C.c_void_p(stream) if stream else None), "umx_energy_forces_dev")

    # ---- graph-parallel single-image mode (reference workers > 1; see parallel.GraphParallelEvaluator) -----------------
    def gp_begin(self, d_pos: int, node_lo: int, node_hi: int, d_energy: int, d_forces: int, stream: int = 0):
        self._chk(self.lib.umx_gp_begin(self._h, C.c_void_p(d_pos), int(node_lo), int(node_hi), C.c_void_p(d_energy), C.c_void_p(d_forces),
                                        C.c_void_p(stream) if stream else None), "umx_gp_begin")

    def gp_step(self) -> Tuple[int, int, bool]:
        """Issue segments up to the next exchange point: (device pointer, float32 count, done)."""
        buf, cnt, done = C.c_void_p(), C.c_size_t(), C.c_int()
        self._chk(self.lib.umx_gp_step(self._h, C.byref(buf), C.byref(cnt), C.byref(done)), "umx_gp_step")
        return int(buf.value or 0), int(cnt.value), bool(done.value)

    def synchronize(self):
        self._chk(self.lib.umx_synchronize(self._h), "umx_synchronize")

    def bond_changes(self, r1: np.ndarray, r2: np.ndarray, cov: np.ndarray, bond_factor: float = 1.20,
                     margin_fraction: float = 0.05, delta_fraction: float = 0.05, distances: bool = True):
        """Pairwise float64 distances of two geometries + formed/broken code matrix (uint8, i<j: 1 formed, 2 broken)."""
        a = np.ascontiguousarray(r1, dtype=np.float64).reshape(-1, 3)
        b = np.ascontiguousarray(r2, dtype=np.float64).reshape(-1, 3)
        c = np.ascontiguousarray(cov, dtype=np.float64).reshape(-1)
        n = a.shape[0]
        if b.shape[0] != n or c.shape[0] != n or n == 0:
            raise ValueError(f"bond_changes: inconsistent sizes {a.shape}, {b.shape}, {c.shape}")
        dp = C.POINTER(C.c_double)
        d1 = np.empty((n, n), dtype=np.float64) if distances else None
        d2 = np.empty((n, n), dtype=np.float64) if distances else None
        code = np.empty((n, n), dtype=np.uint8)
        self._chk(self.lib.umx_bond_changes(self._h, n, a.ctypes.data_as(dp), b.ctypes.data_as(dp), c.ctypes.data_as(dp),
                                            float(bond_factor), float(margin_fraction), float(delta_fraction),
                                            d1.ctypes.data_as(dp) if distances else None, d2.ctypes.data_as(dp) if distances else None,
                                            code.ctypes.data_as(C.POINTER(C.c_uint8))), "umx_bond_changes")
        return d1, d2, code

    # ---- diagnostics -----------------------------------------------------------------------------
    def graph_stats(self) -> Tuple[int, int]:
        ne, md = C.c_int64(), C.c_int32()
        self._chk(self.lib.umx_last_graph_stats(self._h, C.byref(ne), C.byref(md)), "umx_last_graph_stats")
        return int(ne.value), int(md.value)

    def reserve_images(self, n_images: int):
        """Announce batches of up to ``n_images`` images: the next evaluation sizes the workspace once for that many (``umx_reserve_images``)."""
        self._chk(self.lib.umx_reserve_images(self._h, int(n_images)), "umx_reserve_images")

    def workspace_stats(self) -> Tuple[int, int]:
        """(bytes of the HBM workspace, number of times it has been (re-)allocated)."""
        b, n = C.c_int64(), C.c_int32()
        self._chk(self.lib.umx_workspace_stats(self._h, C.byref(b), C.byref(n)), "umx_workspace_stats")
        return int(b.value), int(n.value)

    def last_partitions(self) -> int:
        """Target-node partitions per image of the most recent evaluation (0: the ordinary path; see ``umx_last_partitions``)."""
        return int(self.lib.umx_last_partitions(self._h))

    def last_lanes(self) -> int:
        """Chunks in flight (1 or 2) of the most recent evaluation (``umx_last_lanes``)."""
        return int(self.lib.umx_last_lanes(self._h))

    def profile_enable(self, on: bool = True):
        self._chk(self.lib.umx_profile_enable(self._h, int(on)), "umx_profile_enable")

    def profile_read(self, reset: bool = True):
        st = ProfileStats()
        self._chk(self.lib.umx_profile_read(self._h, C.byref(st), int(reset)), "umx_profile_read")
        fam = [{"ms": st.ms[i], "launches": int(st.launches[i]), "alg_flops": st.alg_flops[i], "mfma_flops": st.mfma_flops[i]} for i in range(3)]
        return {"split_bf16": fam[0], "fp32": fam[1], "radial": fam[2], "gemm_ms": fam[0]["ms"] + fam[1]["ms"], "gemm_launches": fam[0]["launches"] + fam[1]["launches"],
                "gemm_flops": fam[0]["alg_flops"] + fam[1]["alg_flops"]}

    def debug_keep(self, on: bool = True):
        self._chk(self.lib.umx_debug_keep(self._h, int(on)), "umx_debug_keep")

    def debug_fetch(self, name: str, dtype=np.float32) -> np.ndarray:
        nb = C.c_size_t()
        self._chk(self.lib.umx_debug_fetch(self._h, name.encode(), None, 0, C.byref(nb)), "umx_debug_fetch")
        out = np.empty(nb.value // np.dtype(dtype).itemsize, dtype=dtype)
        self._chk(self.lib.umx_debug_fetch(self._h, name.encode(), out.ctypes.data_as(C.c_void_p), out.nbytes, None),
                  "umx_debug_fetch")
        return out
