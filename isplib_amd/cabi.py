"""ctypes view of the C ABI (include/isplib_hip.h) for device tensors.

This is the binding a non-torch host (the reference's own launcher,
csrc/fusedmm.cpp:198) would use; here it lets the GPU parity tests and
``bench.py`` call the library exactly at the drop-in boundary, bypassing the
torch operator layer.  Torch is used only to own device memory and name the
stream.  Every function raises ``RuntimeError`` on a non-zero status.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from . import _lib

# values of include/isplib_hip.h
MSG_SPMM_SUM = 0x2 | 0x00 | 0x100 | 0x1000 | 0x10000
MSG_SPMM_MEAN = 0x2 | 0x00 | 0x100 | 0x3000 | 0x10000
MSG_SPMM_MAX = 0x2 | 0x00 | 0x100 | 0x1000 | 0x20000
MSG_SPMM_MIN = 0x2 | 0x00 | 0x100 | 0x1000 | 0x30000
MESSAGE = {"sum": MSG_SPMM_SUM, "add": MSG_SPMM_SUM, "mean": MSG_SPMM_MEAN, "max": MSG_SPMM_MAX, "min": MSG_SPMM_MIN}

SUCCESS, FAIL, NOT_ENOUGH_MEM, NO_OPT_IMPL, HIP_ERROR = 0, 1, -1, 128, 256
UNDEFINED_USER_FUNCTION = 64

# FusedMM stage flags (csrc/fusedMM.h:18-74) and the built-in SOP_UDEF menu (enum isplib_sop_udef)
VOP = {"copy_lhs": 0x1, "copy_rhs": 0x2, "add": 0x3, "subl": 0x4, "subr": 0x5, "max": 0x6, "min": 0x7, "udef": 0xF}
ROP = {"noop": 0x00, "dot": 0x10, "add_lhs": 0x20, "add_rhs": 0x30, "norml": 0x40, "normr": 0x50, "udef": 0xF0}
SOP = {"noop": 0x000, "copy": 0x100, "udef": 0xF00}
VSC = {"noop": 0x0000, "mul": 0x1000, "add": 0x2000, "mean": 0x3000, "udef": 0xF000}
AOP = {"add": 0x10000, "max": 0x20000, "min": 0x30000, "udef": 0xF0000}
SOP_UDEF = {"none": 0, "sigmoid": 1, "one_minus_sigmoid": 2, "tdist": 3, "scale": 4, "exp": 5, "leaky_exp": 6}
# the FusedMM paper's named patterns as message words (+ the menu entry their SOP_UDEF stands for)
PATTERNS = {
    "spmm": (MSG_SPMM_SUM, "none"),
    "sigmoid_embedding": (VOP["copy_rhs"] | ROP["dot"] | SOP["udef"] | VSC["mul"] | AOP["add"], "sigmoid"),
    "tdist_embedding": (VOP["subr"] | ROP["normr"] | SOP["udef"] | VSC["mul"] | AOP["add"], "tdist"),
    "attention_sum": (VOP["copy_rhs"] | ROP["dot"] | SOP["udef"] | VSC["mul"] | AOP["add"], "leaky_exp"),
}

EXPORTS = (
    "isplib_hip_abi_version", "isplib_hip_last_error", "isplib_hip_set_empty_row", "isplib_hip_get_empty_row", "fusedMM_csr_hip", "performDummySpMM_hip",
    "isplib_spmm_minmax_bw_hip", "isplib_sddmm_csr_hip", "isplib_csr_row_ids_hip",
    "isplib_csr2csc_workspace_bytes", "isplib_csr2csc_hip",
    "isplib_spmm_slices_bytes", "isplib_spmm_slices_build_hip", "isplib_spmm_sliced_workspace_bytes",
    "fusedMM_csr_sliced_hip", "fusedMM_csr_sliced_phase_hip", "isplib_hip_tune",
    "isplib_spmm_tasks_workspace_bytes", "fusedMM_csr_tasks_hip",
    "isplib_spmm_tasks_plan_workspace_bytes", "isplib_spmm_tasks_count_hip", "isplib_spmm_tasks_fill_hip",
    "isplib_sddmm_csr_tasks_hip", "fusedMM_csr_tasks_epilogue_hip", "fusedMM_csr_udef_hip", "fusedMM_csr_udef_tasks_hip", "isplib_pack_indices_hip",
    "isplib_suggest_slices", "isplib_graph_create", "isplib_graph_set_slices", "isplib_graph_set_values", "isplib_graph_spmm", "isplib_graph_spmm_backward",
    "isplib_graph_destroy", "isplib_suggest_slices_whole_rows", "isplib_graph_sddmm",
    "fusedMM_csr_stream_hip", "isplib_spmm_stream_workspace_bytes", "isplib_spmm_stream_geometry", "isplib_suggest_stream", "isplib_suggest_stream_weighted", "isplib_stream_plan_build_hip", "isplib_stream_plan_build_minmax_hip", "fusedMM_csr_stream_minmax_hip", "isplib_spmm_stream_minmax_geometry", "isplib_suggest_stream_minmax", "isplib_spmm_stream_minmax_workspace_bytes", "isplib_stream_plan_set_values_hip", "isplib_stream_plan_free", "isplib_spmm_minmax_bw_det_hip", "isplib_spmm_minmax_bw_workspace_bytes", "isplib_scatter_rows_det_hip",
    "isplib_fusedmm_stream_geometry", "isplib_suggest_fusedmm_stream", "isplib_stream_plan_build_fusedmm_hip", "fusedMM_csr_udef_stream_hip",
    "isplib_row_scale_hip", "isplib_masked_scale_colsum_hip", "isplib_masked_scale_colsum_workspace_bytes",
    "fusedMM_csr_ordered_hip", "isplib_community_order_hip", "isplib_community_order_workspace_bytes", "isplib_order_locality_hip",
    "isplib_graph_set_row_order",
)

# include/isplib_hip_experimental.h (libisplib_hip_exp.so): forms measured slower than the defaults; tests and experiment scripts only
EXP_EXPORTS = ("fusedMM_csr_sweep_hip", "isplib_spmm_sweep_workspace_bytes", "isplib_spmm_sweep_resident_waves",
               "fusedMM_csr_hybrid_hip", "isplib_spmm_hybrid_geometry", "isplib_spmm_hybrid_workspace_bytes", "isplib_sddmm_stream_hip",
               "isplib_hip_tune_experimental")

_i64, _f32, _vp, _i32 = ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_int32


class Epilogue(ctypes.Structure):          # isplib_epilogue
    _fields_ = [("row_scale", ctypes.c_void_p), ("self", ctypes.c_void_p), ("ld_self", ctypes.c_int64),
                ("bias", ctypes.c_void_p), ("relu", ctypes.c_int)]


class TaskPlanInfo(ctypes.Structure):      # isplib_task_plan_info
    _fields_ = [("n_tasks", ctypes.c_int64), ("lane_off", ctypes.c_int64 * 9), ("slices", ctypes.c_int32),
                ("chunk", ctypes.c_int32), ("short_row", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class SweepPlanStruct(ctypes.Structure):   # isplib_sweep_plan
    _fields_ = [("rows", ctypes.c_int64), ("slices", ctypes.c_int32), ("gens", ctypes.c_int32),
                ("waves_per_gen", ctypes.c_int32), ("rows_per_wave", ctypes.c_int32), ("n_tasks", ctypes.c_int64),
                ("n_parts", ctypes.c_int64), ("n_hub", ctypes.c_int64), ("wave_row", ctypes.c_void_p),
                ("wave_part", ctypes.c_void_p), ("wave_task_off", ctypes.c_void_p), ("task_b", ctypes.c_void_p),
                ("task_meta", ctypes.c_void_p), ("hub_row", ctypes.c_void_p), ("hub_off", ctypes.c_void_p)]


class StreamPlanStruct(ctypes.Structure):  # isplib_stream_plan
    _fields_ = [("rows", ctypes.c_int64), ("cols", ctypes.c_int64), ("slices", ctypes.c_int32), ("gens", ctypes.c_int32),
                ("waves_per_gen", ctypes.c_int32), ("rows_per_wave", ctypes.c_int32), ("streams", ctypes.c_int32),
                ("chunk", ctypes.c_int32), ("n_steps", ctypes.c_int64), ("n_parts", ctypes.c_int64), ("n_hub", ctypes.c_int64),
                ("words", ctypes.c_void_p), ("vals", ctypes.c_void_p), ("wave_step_off", ctypes.c_void_p),
                ("wave_row", ctypes.c_void_p), ("wave_part", ctypes.c_void_p),
                ("hub_row", ctypes.c_void_p), ("hub_off", ctypes.c_void_p), ("perm", ctypes.c_void_p)]


class HybridPlanStruct(ctypes.Structure):  # isplib_hybrid_plan
    _fields_ = [("cold", StreamPlanStruct), ("table_rows", ctypes.c_int32), ("hot_cap", ctypes.c_int32), ("n_hot_steps", ctypes.c_int64),
                ("hot_rows", ctypes.c_void_p), ("hot_words", ctypes.c_void_p), ("hot_step_off", ctypes.c_void_p),
                ("hot_perm", ctypes.c_void_p)]


_sigs_set = False


def lib() -> ctypes.CDLL:
    global _sigs_set
    L = _lib.cdll()
    if not _sigs_set:
        L.isplib_hip_abi_version.restype = ctypes.c_int
        L.isplib_hip_last_error.restype = ctypes.c_char_p
        L.fusedMM_csr_hip.restype = ctypes.c_int
        L.fusedMM_csr_hip.argtypes = [_i32, _i64, _i64, _i64, _f32, _i64, _i64, _i64, _vp, _vp, _vp, _vp,
                                      _vp, _i64, _vp, _i64, _f32, _vp, _i64, _vp, _vp]
        L.fusedMM_csr_udef_hip.restype = ctypes.c_int
        L.fusedMM_csr_udef_hip.argtypes = [_i32, _i64, _i64, _i64, _f32, _i64, _i64, _i64, _vp, _vp, _vp, _vp,
                                           _vp, _i64, _vp, _i64, _f32, _vp, _i64, _vp, ctypes.c_int, _f32, _vp]
        L.fusedMM_csr_udef_tasks_hip.restype = ctypes.c_int
        L.fusedMM_csr_udef_tasks_hip.argtypes = [_i32, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp,
                                                 _vp, _vp, ctypes.c_int, _vp, _vp, _i64, _vp, _i64, _vp, ctypes.c_int, _f32,
                                                 _vp, ctypes.c_size_t, _vp]
        L.performDummySpMM_hip.restype = None
        L.performDummySpMM_hip.argtypes = [_i64, _vp]
        L.isplib_spmm_minmax_bw_hip.restype = ctypes.c_int
        L.isplib_spmm_minmax_bw_hip.argtypes = [_i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
        L.isplib_sddmm_csr_hip.restype = ctypes.c_int
        L.isplib_sddmm_csr_hip.argtypes = [_i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _i64, ctypes.c_int, _vp, _vp]
        L.isplib_csr_row_ids_hip.restype = ctypes.c_int
        L.isplib_csr_row_ids_hip.argtypes = [_i64, _i64, _vp, _vp, _vp]
        L.isplib_csr2csc_workspace_bytes.restype = ctypes.c_size_t
        L.isplib_csr2csc_workspace_bytes.argtypes = [_i64, _i64, _i64]
        L.isplib_csr2csc_hip.restype = ctypes.c_int
        L.isplib_csr2csc_hip.argtypes = [_i64, _i64, _i64, _vp, _vp, _vp, ctypes.c_int, _vp, _vp, _vp, _vp,
                                         _vp, ctypes.c_size_t, _vp]
        L.isplib_spmm_slices_bytes.restype = ctypes.c_size_t
        L.isplib_spmm_slices_bytes.argtypes = [_i64, ctypes.c_int]
        L.isplib_spmm_slices_build_hip.restype = ctypes.c_int
        L.isplib_spmm_slices_build_hip.argtypes = [_i64, _i64, _i64, _vp, _vp, _vp, ctypes.c_int, _vp, _vp, _vp]
        L.isplib_spmm_sliced_workspace_bytes.restype = ctypes.c_size_t
        L.isplib_spmm_sliced_workspace_bytes.argtypes = [_i32, _i64, _i64, ctypes.c_int]
        L.fusedMM_csr_sliced_hip.restype = ctypes.c_int
        L.fusedMM_csr_sliced_hip.argtypes = [_i32, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, ctypes.c_int,
                                             _vp, _i64, _vp, _i64, _vp, _vp, ctypes.c_size_t, _vp]
        L.fusedMM_csr_sliced_phase_hip.restype = ctypes.c_int
        L.fusedMM_csr_sliced_phase_hip.argtypes = [_i32, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, ctypes.c_int,
                                                   ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _i64, _vp, _i64, _vp,
                                                   _vp, ctypes.c_size_t, _vp]
        L.isplib_spmm_tasks_workspace_bytes.restype = ctypes.c_size_t
        L.isplib_spmm_tasks_workspace_bytes.argtypes = [_i32, _i64, _i64]
        L.fusedMM_csr_tasks_hip.restype = ctypes.c_int
        L.isplib_pack_indices_hip.restype = ctypes.c_int
        L.isplib_pack_indices_hip.argtypes = [_i64, _vp, _vp, _vp]
        L.fusedMM_csr_tasks_hip.argtypes = [_i32, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp,
                                            ctypes.c_int, _vp, _vp, _i64, _vp, _i64, _vp, _vp, ctypes.c_size_t, _vp]
        L.isplib_spmm_tasks_plan_workspace_bytes.restype = ctypes.c_size_t
        L.isplib_spmm_tasks_plan_workspace_bytes.argtypes = [_i64, ctypes.c_int]
        L.isplib_spmm_tasks_count_hip.restype = ctypes.c_int
        L.isplib_spmm_tasks_count_hip.argtypes = [_i64, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp,
                                                  ctypes.c_size_t, ctypes.POINTER(TaskPlanInfo), _vp]
        L.isplib_spmm_tasks_fill_hip.restype = ctypes.c_int
        L.isplib_spmm_tasks_fill_hip.argtypes = [_i64, _vp, _vp, _vp, ctypes.POINTER(TaskPlanInfo), _vp, _vp, _vp, _vp, _vp]
        L.isplib_sddmm_csr_tasks_hip.restype = ctypes.c_int
        L.isplib_sddmm_csr_tasks_hip.argtypes = [_i64, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp,
                                                 _i64, ctypes.c_int, _vp, _vp]
        L.fusedMM_csr_tasks_epilogue_hip.restype = ctypes.c_int
        L.fusedMM_csr_tasks_epilogue_hip.argtypes = [_i32, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp,
                                                     ctypes.c_int, _vp, _vp, _i64, _vp, _i64, _vp, ctypes.c_size_t,
                                                     ctypes.POINTER(Epilogue), _vp]
        L.isplib_suggest_slices.restype = ctypes.c_int
        L.isplib_suggest_slices.argtypes = [_i64, _i64, _i64, _i64, ctypes.c_int]
        L.isplib_graph_create.restype = ctypes.c_int
        L.isplib_graph_create.argtypes = [_i64, _i64, _i64, _vp, _vp, _vp, ctypes.POINTER(_vp)]
        L.isplib_graph_set_slices.restype = ctypes.c_int
        L.isplib_graph_set_slices.argtypes = [_vp, ctypes.c_int]
        L.isplib_graph_set_values.restype = ctypes.c_int
        L.isplib_graph_set_values.argtypes = [_vp, _vp]
        L.isplib_graph_spmm.restype = ctypes.c_int
        L.isplib_graph_spmm.argtypes = [_vp, _i32, _i64, _vp, _i64, _vp, _i64, _vp, _vp]
        L.isplib_graph_spmm_backward.restype = ctypes.c_int
        L.isplib_graph_spmm_backward.argtypes = [_vp, ctypes.c_int, _i64, _vp, _i64, _vp, _i64, _vp]
        L.isplib_suggest_slices_whole_rows.restype = ctypes.c_int
        L.isplib_suggest_slices_whole_rows.argtypes = [_i64, _i64, _i64, _i64]
        L.isplib_graph_sddmm.restype = ctypes.c_int
        L.isplib_graph_sddmm.argtypes = [_vp, ctypes.c_int, _i64, _vp, _i64, _vp, _i64, _vp, _vp]
        L.isplib_graph_destroy.restype = None
        L.isplib_graph_destroy.argtypes = [_vp]
        L.isplib_hip_tune.restype = ctypes.c_int
        L.isplib_hip_tune.argtypes = [ctypes.c_int, ctypes.c_int]
        L.isplib_spmm_minmax_bw_workspace_bytes.restype = ctypes.c_size_t
        L.isplib_spmm_minmax_bw_workspace_bytes.argtypes = [_i64, _i64, _i64]
        L.isplib_spmm_minmax_bw_det_hip.restype = ctypes.c_int
        L.isplib_spmm_minmax_bw_det_hip.argtypes = [_i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]
        L.isplib_fusedmm_stream_geometry.restype = ctypes.c_int
        L.isplib_fusedmm_stream_geometry.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.isplib_suggest_fusedmm_stream.restype = ctypes.c_int
        L.isplib_suggest_fusedmm_stream.argtypes = [_i32, _i64, _i64, _i64, _i64] + [ctypes.POINTER(ctypes.c_int)] * 3
        L.isplib_stream_plan_build_fusedmm_hip.restype = ctypes.c_int
        L.isplib_stream_plan_build_fusedmm_hip.argtypes = [_i64, _i64, _i64, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                           ctypes.POINTER(StreamPlanStruct), _vp]
        L.fusedMM_csr_udef_stream_hip.restype = ctypes.c_int
        L.fusedMM_csr_udef_stream_hip.argtypes = [_i32, _i64, _i64, _i64, _i64, _vp, _vp, ctypes.POINTER(StreamPlanStruct), _vp, _i64, _vp, _i64,
                                                  _vp, _i64, ctypes.c_int, _f32, _vp, ctypes.c_size_t, _vp]
        L.isplib_row_scale_hip.restype = ctypes.c_int
        L.isplib_row_scale_hip.argtypes = [_i64, _i64, _vp, _i64, _vp, _vp, _i64, _vp]
        L.isplib_masked_scale_colsum_workspace_bytes.restype = ctypes.c_size_t
        L.isplib_masked_scale_colsum_workspace_bytes.argtypes = [_i64, _i64]
        L.isplib_masked_scale_colsum_hip.restype = ctypes.c_int
        L.isplib_masked_scale_colsum_hip.argtypes = [_i64, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, ctypes.c_size_t, _vp]
        L.isplib_scatter_rows_det_hip.restype = ctypes.c_int
        L.isplib_scatter_rows_det_hip.argtypes = [_i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]
        L.isplib_stream_plan_build_hip.restype = ctypes.c_int
        L.isplib_stream_plan_build_hip.argtypes = [_i64, _i64, _i64, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                   ctypes.POINTER(StreamPlanStruct), _vp]
        L.isplib_stream_plan_build_minmax_hip.restype = ctypes.c_int
        L.isplib_stream_plan_build_minmax_hip.argtypes = [_i64, _i64, _i64, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                          ctypes.POINTER(StreamPlanStruct), _vp]
        L.isplib_spmm_stream_minmax_geometry.restype = ctypes.c_int
        L.isplib_spmm_stream_minmax_geometry.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.isplib_suggest_stream_minmax.restype = ctypes.c_int
        L.isplib_suggest_stream_minmax.argtypes = [_i64, _i64, _i64, _i64, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.isplib_spmm_stream_minmax_workspace_bytes.restype = ctypes.c_size_t
        L.isplib_spmm_stream_minmax_workspace_bytes.argtypes = [ctypes.POINTER(StreamPlanStruct)]
        L.fusedMM_csr_stream_minmax_hip.restype = ctypes.c_int
        L.fusedMM_csr_stream_minmax_hip.argtypes = [_i32, _i64, _i64, _i64, _i64, _vp, _vp, ctypes.POINTER(StreamPlanStruct), _vp, _i64,
                                                    _vp, _i64, _vp, _vp, ctypes.c_size_t, _vp]
        L.isplib_stream_plan_set_values_hip.restype = ctypes.c_int
        L.isplib_stream_plan_set_values_hip.argtypes = [ctypes.POINTER(StreamPlanStruct), _vp, _vp]
        L.isplib_stream_plan_free.restype = None
        L.isplib_stream_plan_free.argtypes = [ctypes.POINTER(StreamPlanStruct)]
        L.isplib_community_order_workspace_bytes.restype = ctypes.c_size_t
        L.isplib_community_order_workspace_bytes.argtypes = [_i64, _i64]
        L.isplib_community_order_hip.restype = ctypes.c_int
        L.isplib_community_order_hip.argtypes = [_i64, _i64, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, ctypes.POINTER(ctypes.c_int), _vp,
                                                 ctypes.c_size_t, _vp]
        L.isplib_order_locality_hip.restype = ctypes.c_int
        L.isplib_order_locality_hip.argtypes = [_i64, _i64, _vp, _vp, _vp, _i64, ctypes.POINTER(ctypes.c_double), _vp, ctypes.c_size_t, _vp]
        L.isplib_graph_set_row_order.restype = ctypes.c_int
        L.isplib_graph_set_row_order.argtypes = [_vp, _vp, _vp]
        L.fusedMM_csr_ordered_hip.restype = ctypes.c_int
        L.fusedMM_csr_ordered_hip.argtypes = [_i32, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp]
        L.isplib_suggest_stream.restype = ctypes.c_int
        L.isplib_suggest_stream.argtypes = [_i64, _i64, _i64, _i64, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.isplib_suggest_stream_weighted.restype = ctypes.c_int
        L.isplib_suggest_stream_weighted.argtypes = [_i64, _i64, _i64, _i64, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.isplib_spmm_stream_geometry.restype = ctypes.c_int
        L.isplib_spmm_stream_geometry.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.isplib_spmm_stream_workspace_bytes.restype = ctypes.c_size_t
        L.isplib_spmm_stream_workspace_bytes.argtypes = [ctypes.POINTER(StreamPlanStruct)]
        L.fusedMM_csr_stream_hip.restype = ctypes.c_int
        L.fusedMM_csr_stream_hip.argtypes = [_i32, _i64, _i64, _i64, _i64, _vp, _vp, ctypes.POINTER(StreamPlanStruct), _vp, _i64,
                                             _vp, _i64, _vp, ctypes.c_size_t, ctypes.POINTER(Epilogue), _vp]
        _sigs_set = True
    return L


_exp = None


def exp_lib() -> ctypes.CDLL:
    """libisplib_hip_exp.so (include/isplib_hip_experimental.h): the sweep schedule, the LDS hot-row hybrid and the stream-plan
    SDDMM -- built, bit-exact, measured slower than the defaults, kept for tests and experiments.  Loaded on first use."""
    global _exp
    if _exp is None:
        lib()                                     # the default library first: the experimental one links against it
        path = os.path.join(os.path.dirname(_lib.CABI_PATH), "libisplib_hip_exp.so")
        if not os.path.exists(path):
            raise ImportError(f"isplib_amd: '{os.path.basename(path)}' not found; build it with `make -C isplib_amd/csrc`")
        L = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        L.isplib_hip_tune_experimental.restype = ctypes.c_int
        L.isplib_hip_tune_experimental.argtypes = [ctypes.c_int, ctypes.c_int]
        L.isplib_spmm_hybrid_geometry.restype = ctypes.c_int
        L.isplib_spmm_hybrid_geometry.argtypes = [ctypes.c_int] + [ctypes.POINTER(ctypes.c_int)] * 4
        L.isplib_spmm_hybrid_workspace_bytes.restype = ctypes.c_size_t
        L.isplib_spmm_hybrid_workspace_bytes.argtypes = [ctypes.POINTER(HybridPlanStruct)]
        L.fusedMM_csr_hybrid_hip.restype = ctypes.c_int
        L.fusedMM_csr_hybrid_hip.argtypes = [_i32, _i64, _i64, _i64, _i64, _vp, _vp, ctypes.POINTER(HybridPlanStruct), _vp, _i64, _vp, _i64,
                                             _vp, ctypes.c_size_t, _vp, _vp]
        L.isplib_sddmm_stream_hip.restype = ctypes.c_int
        L.isplib_sddmm_stream_hip.argtypes = [_i64, _i64, _i64, _i64, _vp, _vp, ctypes.POINTER(StreamPlanStruct), _vp, _i64, _vp, _i64,
                                              ctypes.c_int, _vp, _vp]
        L.isplib_spmm_sweep_resident_waves.restype = ctypes.c_int
        L.isplib_spmm_sweep_resident_waves.argtypes = [_i32, _i64, ctypes.c_int]
        L.isplib_spmm_sweep_workspace_bytes.restype = ctypes.c_size_t
        L.isplib_spmm_sweep_workspace_bytes.argtypes = [_i32, ctypes.POINTER(SweepPlanStruct), _i64]
        L.fusedMM_csr_sweep_hip.restype = ctypes.c_int
        L.fusedMM_csr_sweep_hip.argtypes = [_i32, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(SweepPlanStruct),
                                            _vp, _i64, _vp, _i64, _vp, _vp, ctypes.c_size_t, ctypes.POINTER(Epilogue), _vp]
        _exp = L
    return _exp


def last_error() -> str:
    return lib().isplib_hip_last_error().decode()


# status codes of include/isplib_hip.h (csrc/fusedMM.h:105-114 of the reference)
ISPLIB_SUCCESS, ISPLIB_FAIL, ISPLIB_NOT_ENOUGH_MEM, ISPLIB_NO_OPT_IMPL, ISPLIB_HIP_ERROR = 0, 1, -1, 128, 256


class IsplibError(RuntimeError):
    """A non-zero status of the C ABI (include/isplib_hip.h: ISPLIB_*), with the library's message."""

    def __init__(self, status: int, what: str, message: str):
        super().__init__(f"{what} failed with status {status}: {message}")
        self.status = int(status)


def _check(status: int, what: str) -> None:
    if status != 0:
        raise IsplibError(status, what, last_error())


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device) -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _dev(t: torch.Tensor, name: str, dtype) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"isplib_amd: `{name}` must be a GPU tensor -- there is no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"isplib_amd: `{name}` must be {dtype}, got {t.dtype}")
    return t.contiguous()


def fusedMM_csr_hip(imessage: int, rowptr: torch.Tensor, col: torch.Tensor, val: Optional[torch.Tensor],
                    y: torch.Tensor, z: torch.Tensor, z_arg: Optional[torch.Tensor] = None, *, beta: float = 0.0,
                    check: bool = True) -> int:
    """Raw boundary call with the reference's argument pattern (csrc/fusedmm.cpp:198):
    pntrb = rowptr, pntre = rowptr + 1, ldy/ldz = row strides of y/z."""
    rowptr = _dev(rowptr, "rowptr", torch.int64)
    col = _dev(col, "col", torch.int64)
    if val is not None:
        val = _dev(val, "val", torch.float32)
    assert y.is_cuda and z.is_cuda and y.dtype == torch.float32 and z.dtype == torch.float32
    assert y.dim() == 2 and z.dim() == 2 and y.stride(1) == 1 and z.stride(1) == 1
    m, n, k = rowptr.numel() - 1, y.size(0), y.size(1)
    assert z.size(0) == m and z.size(1) == k
    rp = rowptr.data_ptr()
    with torch.cuda.device(y.device):
        st = lib().fusedMM_csr_hip(int(imessage), m, n, k, 1.0, col.numel(), m, n, _ptr(val), _ptr(col),
                                   ctypes.c_void_p(rp), ctypes.c_void_p(rp + 8), None, k, _ptr(y),
                                   y.stride(0) if n > 1 else max(k, y.stride(0)), beta, _ptr(z),
                                   z.stride(0) if m > 1 else max(k, z.stride(0)), _ptr(z_arg), _stream(y.device))
    if check:
        _check(st, "fusedMM_csr_hip")
    return st


def community_order(rowptr, col, rounds: int = 8, seed: int = 0):
    """(order int32 [m], labels int32 [m], rounds run) by the library's label propagation (isplib_community_order_hip)."""
    rowptr = _dev(rowptr, "rowptr", torch.int64)
    col = _dev(col, "col", torch.int64)
    m, nnz = rowptr.numel() - 1, col.numel()
    order = torch.empty(m, dtype=torch.int32, device=col.device)
    labels = torch.empty(m, dtype=torch.int32, device=col.device)
    ran = ctypes.c_int(0)
    with torch.cuda.device(col.device):
        ws = torch.empty(lib().isplib_community_order_workspace_bytes(m, nnz), dtype=torch.uint8, device=col.device)
        _check(lib().isplib_community_order_hip(m, nnz, _ptr(rowptr), _ptr(col), int(rounds), int(seed), _ptr(order), _ptr(labels),
                                                ctypes.byref(ran), _ptr(ws), ws.numel(), _stream(col.device)), "isplib_community_order_hip")
    return order, labels, ran.value


def order_locality(rowptr, col, order, window: int = 1024) -> float:
    """Share of the stored entries whose column lies within `window` positions of its row in `order` (None: index order)."""
    rowptr = _dev(rowptr, "rowptr", torch.int64)
    col = _dev(col, "col", torch.int64)
    m = rowptr.numel() - 1
    share = ctypes.c_double(0.0)
    with torch.cuda.device(col.device):
        ws = torch.empty(4 * m + 1024, dtype=torch.uint8, device=col.device)
        _check(lib().isplib_order_locality_hip(m, col.numel(), _ptr(rowptr), _ptr(col), _ptr(order), int(window), ctypes.byref(share),
                                               _ptr(ws), ws.numel(), _stream(col.device)), "isplib_order_locality_hip")
    return share.value


def fusedMM_csr_ordered_hip(imessage: int, rowptr, col, val, order, y, z, z_arg=None, check: bool = True) -> int:
    """The plain kernel with the rows taken in `order` (int32 [m], position -> row; None = fusedMM_csr_hip)."""
    rowptr = _dev(rowptr, "rowptr", torch.int64)
    col = _dev(col, "col", torch.int64)
    if order is not None:
        order = _dev(order, "order", torch.int32)
        assert order.numel() == rowptr.numel() - 1
    assert y.is_cuda and z.is_cuda and y.dtype == torch.float32 and z.dtype == torch.float32 and y.stride(1) == 1 and z.stride(1) == 1
    m, n, k = rowptr.numel() - 1, y.size(0), y.size(1)
    rp = rowptr.data_ptr()
    with torch.cuda.device(y.device):
        st = lib().fusedMM_csr_ordered_hip(int(imessage), m, n, k, col.numel(), _ptr(val), _ptr(col), ctypes.c_void_p(rp),
                                           ctypes.c_void_p(rp + 8), _ptr(order), _ptr(y), y.stride(0) if n > 1 else max(k, y.stride(0)),
                                           _ptr(z), z.stride(0) if m > 1 else max(k, z.stride(0)), _ptr(z_arg), _stream(y.device))
    if check:
        _check(st, "fusedMM_csr_ordered_hip")
    return st


def spmm_ordered(rowptr, col, val, order, y, reduce: str = "sum"):
    """Allocate the outputs and call the ordered plain kernel; returns (out, arg|None)."""
    y = y.contiguous()
    m, k = rowptr.numel() - 1, y.size(1)
    out = torch.empty((m, k), dtype=torch.float32, device=y.device)
    arg = torch.empty((m, k), dtype=torch.int64, device=y.device) if reduce in ("max", "min") else None
    fusedMM_csr_ordered_hip(MESSAGE[reduce], rowptr, col, val, order, y, out, arg)
    return out, arg


def fusedmm(imessage: int, rowptr, col, val, x, y, sop_udef="none", sop_param: float = 0.0, check: bool = True, plan=None):
    """The generic FusedMM pipeline (fusedMM_csr_udef_hip): z[i,:] = AOP_j VSC(SOP(ROP(VOP(x_i, y_j))), .) over the
    stored entries of row i.  `imessage` is a word built from VOP/ROP/SOP/VSC/AOP (or PATTERNS[name][0]);
    `sop_udef` names the built-in function a SOP_UDEF stage stands for.  With `plan` (isplib_amd.plan.TaskPlan) the
    task form runs (fusedMM_csr_udef_tasks_hip).  Returns (status, z, z_arg | None)."""
    rowptr = _dev(rowptr, "rowptr", torch.int64)
    col = _dev(col, "col", torch.int64)
    if val is not None:
        val = _dev(val, "val", torch.float32)
    y = _dev(y, "y", torch.float32)
    if x is not None:
        x = _dev(x, "x", torch.float32)
    m, n, k = rowptr.numel() - 1, y.size(0), y.size(1)
    z = torch.empty((m, k), dtype=torch.float32, device=y.device)
    arg = torch.empty((m, k), dtype=torch.int64, device=y.device) if ((imessage >> 16) & 0xF) in (2, 3) else None
    rp = rowptr.data_ptr()
    kind = SOP_UDEF[sop_udef] if isinstance(sop_udef, str) else int(sop_udef)
    with torch.cuda.device(y.device):
        if plan is not None:
            reduce = {1: "sum", 2: "max", 3: "min"}.get((imessage >> 16) & 0xF, "sum")
            work = plan.workspace(reduce, k)
            lane = (ctypes.c_int64 * 9)(*plan.lane_off)
            st = lib().fusedMM_csr_udef_tasks_hip(int(imessage), m, n, k, col.numel(), _ptr(val), _ptr(col), _plan_col32(plan, col),
                                                  ctypes.c_void_p(rp), ctypes.c_void_p(rp + 8), _ptr(x), k, plan.n_tasks,
                                                  _ptr(plan.task_row), _ptr(plan.task_b), _ptr(plan.task_len), _ptr(plan.seg_off),
                                                  plan.slices, lane, _ptr(y), k, _ptr(z), k, _ptr(arg), kind, float(sop_param),
                                                  _ptr(work), work.numel(), _stream(y.device))
        else:
            st = lib().fusedMM_csr_udef_hip(int(imessage), m, n, k, 1.0, col.numel(), m, n, _ptr(val), _ptr(col),
                                            ctypes.c_void_p(rp), ctypes.c_void_p(rp + 8), _ptr(x), k, _ptr(y), k, 0.0,
                                            _ptr(z), k, _ptr(arg), kind, float(sop_param), _stream(y.device))
    if check:
        _check(st, "fusedMM_csr_udef_tasks_hip" if plan is not None else "fusedMM_csr_udef_hip")
    return st, z, arg


def suggest_fusedmm_stream(imessage: int, m: int, n: int, nnz: int, k: int):
    """(streams, slices, chunk) when the word should run on the stream front end (isplib_suggest_fusedmm_stream), else None."""
    v = [ctypes.c_int(0) for _ in range(3)]
    if not lib().isplib_suggest_fusedmm_stream(int(imessage), int(m), int(n), int(nnz), int(k), *[ctypes.byref(x) for x in v]):
        return None
    return tuple(int(x.value) for x in v)


def fusedmm_stream_geometry(streams: int = 2):
    rpw, res = ctypes.c_int(0), ctypes.c_int(0)
    _check(lib().isplib_fusedmm_stream_geometry(int(streams), ctypes.byref(rpw), ctypes.byref(res)), "isplib_fusedmm_stream_geometry")
    return int(rpw.value), int(res.value)


def fusedmm_stream(imessage: int, rowptr, nnz: int, plan, x, y, sop_udef="none", sop_param: float = 0.0, check: bool = True):
    """The two SDDMM-fused words on the stream front end (fusedMM_csr_udef_stream_hip); `plan`: a NativeStreamPlan built with
    fusedmm=True.  Returns (status, z)."""
    rowptr = _dev(rowptr, "rowptr", torch.int64)
    x, y = _dev(x, "x", torch.float32), _dev(y, "y", torch.float32)
    m, n, k = rowptr.numel() - 1, y.size(0), y.size(1)
    z = torch.empty((m, k), dtype=torch.float32, device=y.device)
    kind = SOP_UDEF[sop_udef] if isinstance(sop_udef, str) else int(sop_udef)
    work = plan.workspace()
    rp = rowptr.data_ptr()
    ps = plan.struct()
    with torch.cuda.device(y.device):
        st = lib().fusedMM_csr_udef_stream_hip(int(imessage), m, n, k, int(nnz), ctypes.c_void_p(rp), ctypes.c_void_p(rp + 8), ctypes.byref(ps),
                                               _ptr(x), k, _ptr(y), k, _ptr(z), k, kind, float(sop_param), _ptr(work), work.numel(),
                                               _stream(y.device))
    if check:
        _check(st, "fusedMM_csr_udef_stream_hip")
    return st, z


def spmm(rowptr, col, val, y, reduce: str = "sum"):
    """Allocate outputs and call the boundary; returns (out, arg|None)."""
    m, k = rowptr.numel() - 1, y.size(1)
    y = y.contiguous()
    out = torch.empty((m, k), dtype=torch.float32, device=y.device)
    arg = torch.empty((m, k), dtype=torch.int64, device=y.device) if reduce in ("max", "min") else None
    fusedMM_csr_hip(MESSAGE[reduce], rowptr, col, val, y, out, arg)
    return out, arg


def perform_dummy_spmm(flag: int = 0) -> None:
    lib().performDummySpMM_hip(int(flag), _stream(None))


def spmm_minmax_bw(col, val, mat, arg, grad_out, need_mat=True, need_val=True, deterministic=False):
    """(grad_val, grad_mat) of SpMM-max/min; deterministic=True: the atomic-free form (isplib_spmm_minmax_bw_det_hip)."""
    col = _dev(col, "col", torch.int64)
    mat = _dev(mat, "mat", torch.float32)
    arg = _dev(arg, "arg", torch.int64)
    grad_out = _dev(grad_out, "grad_out", torch.float32)
    if val is not None:
        val = _dev(val, "val", torch.float32)
    m, k = arg.shape
    n, nnz = mat.size(0), col.numel()
    grad_mat = torch.empty_like(mat) if need_mat else None
    grad_val = torch.empty(nnz, dtype=torch.float32, device=mat.device) if need_val else None
    with torch.cuda.device(mat.device):
        if deterministic:
            ws = lib().isplib_spmm_minmax_bw_workspace_bytes(m, n, k)
            work = torch.empty(max(ws, 256), dtype=torch.uint8, device=mat.device)
            st = lib().isplib_spmm_minmax_bw_det_hip(m, n, k, nnz, _ptr(col), _ptr(val), _ptr(mat), _ptr(arg), _ptr(grad_out),
                                                     _ptr(grad_mat), _ptr(grad_val), _ptr(work), work.numel(), _stream(mat.device))
            _check(st, "isplib_spmm_minmax_bw_det_hip")
            return grad_val, grad_mat
        st = lib().isplib_spmm_minmax_bw_hip(m, n, k, nnz, _ptr(col), _ptr(val), _ptr(mat), _ptr(arg), _ptr(grad_out),
                                             _ptr(grad_mat), _ptr(grad_val), _stream(mat.device))
    _check(st, "isplib_spmm_minmax_bw_hip")
    return grad_val, grad_mat


def row_scale(x: torch.Tensor, scale: torch.Tensor, pitch: Optional[int] = None) -> torch.Tensor:
    """y = scale[:, None] * x through isplib_row_scale_hip; `pitch` > k: a [n, k] view of an [n, pitch] buffer (padding 0)."""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    scale = _dev(scale, "scale", torch.float32)
    n, k = x.shape
    ld = k if pitch is None else int(pitch)
    buf = torch.empty((n, ld), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().isplib_row_scale_hip(n, k, _ptr(x), x.stride(0) if n > 1 else max(k, x.stride(0)), _ptr(scale), _ptr(buf), ld,
                                          _stream(x.device)), "isplib_row_scale_hip")
    return buf[:, :k]


def masked_scale_colsum(dz: torch.Tensor, out: Optional[torch.Tensor], scale: Optional[torch.Tensor], want_gy: bool = True,
                        want_bias: bool = True, pitch: Optional[int] = None):
    """(gy, grad_bias) of isplib_masked_scale_colsum_hip: g = dz * (out > 0), gy = g * scale[:, None], grad_bias = g.sum(0)."""
    dz = _dev(dz, "dz", torch.float32)
    out = None if out is None else _dev(out, "out", torch.float32)
    scale = None if scale is None else _dev(scale, "scale", torch.float32)
    n, k = dz.shape
    ld = k if pitch is None else int(pitch)
    gy = torch.empty((n, ld), dtype=torch.float32, device=dz.device) if want_gy else None
    gb = torch.empty(k, dtype=torch.float32, device=dz.device) if want_bias else None
    with torch.cuda.device(dz.device):
        ws = lib().isplib_masked_scale_colsum_workspace_bytes(n, k)
        work = torch.empty(max(ws, 256), dtype=torch.uint8, device=dz.device)
        _check(lib().isplib_masked_scale_colsum_hip(n, k, _ptr(dz), k, _ptr(out), k, _ptr(scale), _ptr(gy), ld, _ptr(gb), _ptr(work),
                                                    work.numel(), _stream(dz.device)), "isplib_masked_scale_colsum_hip")
    return (None if gy is None else gy[:, :k]), gb


def set_empty_row(mode: str = "zero") -> None:
    """What an EMPTY row of max / min holds from now on, every schedule: "zero" (default) or "init" (-FLT_MAX / +FLT_MAX, the
    reference launcher's pre-fill left untouched; include/isplib_hip.h: isplib_hip_set_empty_row).  Positions stay nnz."""
    if mode not in ("zero", "init"):
        raise ValueError("empty-row mode: 'zero' or 'init'")
    _check(lib().isplib_hip_set_empty_row(1 if mode == "init" else 0), "isplib_hip_set_empty_row")


def get_empty_row() -> str:
    return "init" if lib().isplib_hip_get_empty_row() else "zero"


def scatter_rows_det(dest: torch.Tensor, gval: torch.Tensor, lo: int, n: int) -> torch.Tensor:
    """grad[d - lo, c] = sum over i (ascending) of gval[i, c] where dest[i, c] == d, for d in [lo, lo + n)
    (isplib_scatter_rows_det_hip: the local half of the row-partitioned max / min backward)."""
    dest = _dev(dest, "dest", torch.int32)
    gval = _dev(gval, "gval", torch.float32)
    assert dest.shape == gval.shape and dest.dim() == 2
    m, k = dest.shape
    out = torch.empty((n, k), dtype=torch.float32, device=gval.device)
    with torch.cuda.device(gval.device):
        ws = lib().isplib_spmm_minmax_bw_workspace_bytes(m, n, k)
        if ws == 0 and m * k > 0 and n > 0:
            # beyond the sort's 32-bit keys (n * k or m * k >= 2^32): torch's atomic scatter on the same pairs -- correct, not
            # bitwise reproducible (the one place of the partitioned backward that is not; no shape of BASELINE.json gets here)
            d = dest.to(torch.int64) - int(lo)
            mine = (dest >= 0) & (d >= 0) & (d < n)
            rows, cols = mine.nonzero(as_tuple=True)
            out.zero_()
            out.index_put_((d[rows, cols], cols), gval[rows, cols], accumulate=True)
            return out
        work = torch.empty(max(ws, 256), dtype=torch.uint8, device=gval.device)
        _check(lib().isplib_scatter_rows_det_hip(m, n, k, int(lo), _ptr(dest), _ptr(gval), _ptr(out), _ptr(work), work.numel(),
                                                 _stream(gval.device)), "isplib_scatter_rows_det_hip")
    return out


def sddmm(rowptr, col, y, g, mean: bool = False):
    rowptr = _dev(rowptr, "rowptr", torch.int64)
    col = _dev(col, "col", torch.int64)
    y = _dev(y, "y", torch.float32)
    g = _dev(g, "g", torch.float32)
    m, k = rowptr.numel() - 1, y.size(1)
    dval = torch.empty(col.numel(), dtype=torch.float32, device=y.device)
    rp = rowptr.data_ptr()
    with torch.cuda.device(y.device):
        st = lib().isplib_sddmm_csr_hip(m, k, _ptr(col), ctypes.c_void_p(rp), ctypes.c_void_p(rp + 8), _ptr(y), k,
                                        _ptr(g), k, int(bool(mean)), _ptr(dval), _stream(y.device))
    _check(st, "isplib_sddmm_csr_hip")
    return dval


def csr_row_ids(rowptr, nnz: int):
    rowptr = _dev(rowptr, "rowptr", torch.int64)
    row = torch.empty(nnz, dtype=torch.int64, device=rowptr.device)
    with torch.cuda.device(rowptr.device):
        st = lib().isplib_csr_row_ids_hip(rowptr.numel() - 1, nnz, _ptr(rowptr), _ptr(row), _stream(rowptr.device))
    _check(st, "isplib_csr_row_ids_hip")
    return row


def csr2csc(rowptr, col, val, ncols: int, *, mean_scale: bool = False, want_perm: bool = True,
            want_row: bool = True, want_val: bool = True):
    """Device CSR -> CSC operands: (colptr, csr2csc|None, row_t|None, val_t|None)."""
    rowptr = _dev(rowptr, "rowptr", torch.int64)
    col = _dev(col, "col", torch.int64)
    if val is not None:
        val = _dev(val, "val", torch.float32)
    dev = col.device
    m, nnz = rowptr.numel() - 1, col.numel()
    colptr = torch.empty(ncols + 1, dtype=torch.int64, device=dev)
    perm = torch.empty(nnz, dtype=torch.int64, device=dev) if want_perm else None
    row_t = torch.empty(nnz, dtype=torch.int64, device=dev) if want_row else None
    val_t = torch.empty(nnz, dtype=torch.float32, device=dev) if want_val else None
    with torch.cuda.device(dev):
        ws = lib().isplib_csr2csc_workspace_bytes(m, ncols, nnz)
        if ws == 0:
            raise RuntimeError("isplib_csr2csc_workspace_bytes failed: " + last_error())
        work = torch.empty(ws, dtype=torch.uint8, device=dev)
        st = lib().isplib_csr2csc_hip(m, ncols, nnz, _ptr(rowptr), _ptr(col), _ptr(val), int(bool(mean_scale)),
                                      _ptr(colptr), _ptr(perm), _ptr(row_t), _ptr(val_t), _ptr(work), ws, _stream(dev))
    _check(st, "isplib_csr2csc_hip")
    return colptr, perm, row_t, val_t


def spmm_slices(rowptr, col, ncols: int, slices: int = 8, check_sorted: bool = True):
    """Per-graph slice table for fusedMM_csr_sliced_hip: (sliceptr[m*(slices+1)], rows_sorted).
    ``rows_sorted`` is False when some row's columns are not ascending (table unusable);
    reading it synchronises once -- this runs once per graph, never per SpMM."""
    rowptr = _dev(rowptr, "rowptr", torch.int64)
    col = _dev(col, "col", torch.int64)
    dev = col.device
    m = rowptr.numel() - 1
    sliceptr = torch.empty(m * (slices + 1), dtype=torch.int64, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev) if check_sorted else None
    rp = rowptr.data_ptr()
    with torch.cuda.device(dev):
        st = lib().isplib_spmm_slices_build_hip(m, ncols, col.numel(), ctypes.c_void_p(rp), ctypes.c_void_p(rp + 8),
                                                _ptr(col), slices, _ptr(sliceptr), _ptr(flag), _stream(dev))
    _check(st, "isplib_spmm_slices_build_hip")
    return sliceptr, (True if flag is None else int(flag.item()) == 0)


def sliced_workspace(reduce: str, m: int, k: int, slices: int, device) -> torch.Tensor:
    nbytes = lib().isplib_spmm_sliced_workspace_bytes(MESSAGE[reduce], m, k, slices)
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def fusedMM_csr_sliced_hip(imessage: int, rowptr, col, val, sliceptr, slices: int, y, z, z_arg, workspace,
                           check: bool = True) -> int:
    """Raw boundary call of the column-sliced SpMM (operands as fusedMM_csr_hip + slice table + workspace)."""
    assert y.is_cuda and y.dtype == torch.float32 and y.dim() == 2 and y.stride(1) == 1
    m, n, k = rowptr.numel() - 1, y.size(0), y.size(1)
    rp = rowptr.data_ptr()
    with torch.cuda.device(y.device):
        st = lib().fusedMM_csr_sliced_hip(int(imessage), m, n, k, col.numel(), _ptr(val), _ptr(col),
                                          ctypes.c_void_p(rp), ctypes.c_void_p(rp + 8), _ptr(sliceptr), slices,
                                          _ptr(y), y.stride(0) if n > 1 else max(k, y.stride(0)), _ptr(z),
                                          z.stride(0) if m > 1 else max(k, z.stride(0)), _ptr(z_arg),
                                          _ptr(workspace), workspace.numel(), _stream(y.device))
    if check:
        _check(st, "fusedMM_csr_sliced_hip")
    return st


def spmm_sliced(rowptr, col, val, sliceptr, slices: int, y, reduce: str = "sum", workspace=None):
    """Allocate outputs (+ workspace) and call the sliced boundary; returns (out, arg|None)."""
    rowptr = _dev(rowptr, "rowptr", torch.int64)
    col = _dev(col, "col", torch.int64)
    if val is not None:
        val = _dev(val, "val", torch.float32)
    y = y.contiguous()
    m, k = rowptr.numel() - 1, y.size(1)
    out = torch.empty((m, k), dtype=torch.float32, device=y.device)
    arg = torch.empty((m, k), dtype=torch.int64, device=y.device) if reduce in ("max", "min") else None
    if workspace is None:
        workspace = sliced_workspace(reduce, m, k, slices, y.device)
    fusedMM_csr_sliced_hip(MESSAGE[reduce], rowptr, col, val, sliceptr, slices, y, out, arg, workspace)
    return out, arg


def fusedMM_csr_sliced_phase_hip(imessage: int, rowptr, col, val, sliceptr, slices: int, slice_first: int,
                                 slice_count: int, combine: bool, y_ptr: int, n: int, k: int, ldy: int, z, z_arg,
                                 workspace, check: bool = True) -> int:
    """One phase of the column-sliced SpMM (include/isplib_hip.h).  ``y_ptr`` is a raw device address so
    that a phase can read its columns through a base shifted onto a shard of the dense operand."""
    m = rowptr.numel() - 1
    rp = rowptr.data_ptr()
    with torch.cuda.device(z.device):
        st = lib().fusedMM_csr_sliced_phase_hip(
            int(imessage), m, n, k, col.numel(), _ptr(val), _ptr(col), ctypes.c_void_p(rp), ctypes.c_void_p(rp + 8),
            _ptr(sliceptr), slices, slice_first, slice_count, int(bool(combine)), ctypes.c_void_p(y_ptr), ldy, _ptr(z),
            z.stride(0) if m > 1 else max(k, z.stride(0)), _ptr(z_arg), _ptr(workspace), workspace.numel(),
            _stream(z.device))
    if check:
        _check(st, "fusedMM_csr_sliced_phase_hip")
    return st


def pack_indices(col: torch.Tensor) -> torch.Tensor:
    """int32 copy of the column ids (isplib_pack_indices_hip): what the task kernels stream instead of the int64 array."""
    col = _dev(col, "col", torch.int64)
    out = torch.empty(col.numel(), dtype=torch.int32, device=col.device)
    with torch.cuda.device(col.device):
        _check(lib().isplib_pack_indices_hip(col.numel(), _ptr(col), _ptr(out), _stream(col.device)), "isplib_pack_indices_hip")
    return out


def _plan_col32(plan, col):
    """The plan's packed column ids if it has them and they belong to this `col` (same length), else NULL."""
    c32 = getattr(plan, "col32", None)
    return _ptr(c32) if c32 is not None and c32.numel() == col.numel() and c32.device == col.device else None


def fusedMM_csr_tasks_hip(imessage: int, rowptr, col, val, plan, y, z, z_arg, workspace, check: bool = True) -> int:
    """Raw boundary call of the task-list SpMM; ``plan`` is an isplib_amd.plan.TaskPlan."""
    assert y.is_cuda and y.dtype == torch.float32 and y.dim() == 2 and y.stride(1) == 1
    m, n, k = rowptr.numel() - 1, y.size(0), y.size(1)
    rp = rowptr.data_ptr()
    lane = (ctypes.c_int64 * 9)(*plan.lane_off)
    with torch.cuda.device(y.device):
        st = lib().fusedMM_csr_tasks_hip(int(imessage), m, n, k, col.numel(), _ptr(val), _ptr(col), _plan_col32(plan, col), ctypes.c_void_p(rp),
                                         ctypes.c_void_p(rp + 8), plan.n_tasks, _ptr(plan.task_row), _ptr(plan.task_b),
                                         _ptr(plan.task_len), _ptr(plan.seg_off), plan.slices, lane, _ptr(y),
                                         y.stride(0) if n > 1 else max(k, y.stride(0)), _ptr(z),
                                         z.stride(0) if m > 1 else max(k, z.stride(0)), _ptr(z_arg), _ptr(workspace),
                                         workspace.numel(), _stream(y.device))
    if check:
        _check(st, "fusedMM_csr_tasks_hip")
    return st


def spmm_tasks(rowptr, col, val, plan, y, reduce: str = "sum", workspace=None):
    """Allocate outputs (+ workspace) and call the task-list boundary; returns (out, arg|None)."""
    rowptr = _dev(rowptr, "rowptr", torch.int64)
    col = _dev(col, "col", torch.int64)
    if val is not None:
        val = _dev(val, "val", torch.float32)
    y = y.contiguous()
    m, k = rowptr.numel() - 1, y.size(1)
    out = torch.empty((m, k), dtype=torch.float32, device=y.device)
    arg = torch.empty((m, k), dtype=torch.int64, device=y.device) if reduce in ("max", "min") else None
    if workspace is None:
        workspace = plan.workspace(reduce, k)
    fusedMM_csr_tasks_hip(MESSAGE[reduce], rowptr, col, val, plan, y, out, arg, workspace)
    return out, arg


def sddmm_stream(rowptr, nnz: int, plan, y, g, mean: bool = False):
    """dA over the stream plan of the SpMM (isplib_sddmm_stream_hip): dval[e] = <y[col[e]], g[row(e)]> (/ max(deg, 1))."""
    rowptr = _dev(rowptr, "rowptr", torch.int64)
    y, g = y.contiguous(), g.contiguous()
    m, n, k = rowptr.numel() - 1, y.size(0), y.size(1)
    assert g.size(0) == m and g.size(1) == k
    dval = torch.empty(int(nnz), dtype=torch.float32, device=y.device)
    rp = rowptr.data_ptr()
    ps = plan.struct()
    with torch.cuda.device(y.device):
        st = exp_lib().isplib_sddmm_stream_hip(m, n, k, int(nnz), ctypes.c_void_p(rp), ctypes.c_void_p(rp + 8), ctypes.byref(ps), _ptr(y), k,
                                           _ptr(g), k, int(bool(mean)), _ptr(dval), _stream(y.device))
    _check(st, "isplib_sddmm_stream_hip")
    return dval


def sddmm_tasks(rowptr, col, plan, y, g, mean: bool = False):
    """dA over the task plan of the SpMM (isplib_sddmm_csr_tasks_hip)."""
    rowptr = _dev(rowptr, "rowptr", torch.int64)
    col = _dev(col, "col", torch.int64)
    y = _dev(y, "y", torch.float32)
    g = _dev(g, "g", torch.float32)
    m, n, k = rowptr.numel() - 1, y.size(0), y.size(1)
    dval = torch.empty(col.numel(), dtype=torch.float32, device=y.device)
    rp = rowptr.data_ptr()
    lane = (ctypes.c_int64 * 9)(*plan.lane_off)
    with torch.cuda.device(y.device):
        st = lib().isplib_sddmm_csr_tasks_hip(m, n, k, _ptr(col), _plan_col32(plan, col), ctypes.c_void_p(rp), ctypes.c_void_p(rp + 8), plan.n_tasks,
                                              _ptr(plan.task_row), _ptr(plan.task_b), _ptr(plan.task_len), lane, _ptr(y), k,
                                              _ptr(g), k, int(bool(mean)), _ptr(dval), _stream(y.device))
    _check(st, "isplib_sddmm_csr_tasks_hip")
    return dval


def spmm_tasks_epilogue(rowptr, col, val, plan, y, reduce="sum", row_scale=None, self_term=None, bias=None, relu=False):
    """fusedMM_csr_tasks_epilogue_hip: out = act(row_scale * (reduce + self) + bias), sum / mean only."""
    rowptr = _dev(rowptr, "rowptr", torch.int64)
    col = _dev(col, "col", torch.int64)
    y = y.contiguous()
    m, n, k = rowptr.numel() - 1, y.size(0), y.size(1)
    out = torch.empty((m, k), dtype=torch.float32, device=y.device)
    work = plan.workspace(reduce, k)
    ep = Epilogue(None if row_scale is None else row_scale.data_ptr(), None if self_term is None else self_term.data_ptr(),
                  k if self_term is None else self_term.stride(0), None if bias is None else bias.data_ptr(), int(bool(relu)))
    rp = rowptr.data_ptr()
    lane = (ctypes.c_int64 * 9)(*plan.lane_off)
    with torch.cuda.device(y.device):
        st = lib().fusedMM_csr_tasks_epilogue_hip(MESSAGE[reduce], m, n, k, col.numel(), _ptr(val), _ptr(col), _plan_col32(plan, col), ctypes.c_void_p(rp),
                                                  ctypes.c_void_p(rp + 8), plan.n_tasks, _ptr(plan.task_row), _ptr(plan.task_b),
                                                  _ptr(plan.task_len), _ptr(plan.seg_off), plan.slices, lane, _ptr(y), k, _ptr(out),
                                                  k, _ptr(work), work.numel(), ctypes.byref(ep), _stream(y.device))
    _check(st, "fusedMM_csr_tasks_epilogue_hip")
    return out


def fusedMM_csr_sweep_hip(imessage: int, rowptr, col, val, plan, y, z, z_arg, workspace=None, epilogue=None, check: bool = True) -> int:
    """Raw boundary call of the sweep-schedule SpMM; ``plan`` is an isplib_amd.plan.SweepPlan."""
    assert y.is_cuda and y.dtype == torch.float32 and y.dim() == 2 and y.stride(1) == 1
    m, n, k = rowptr.numel() - 1, y.size(0), y.size(1)
    rp = rowptr.data_ptr()
    ps = plan.struct()
    with torch.cuda.device(y.device):
        st = exp_lib().fusedMM_csr_sweep_hip(int(imessage), m, n, k, col.numel(), _ptr(val), _ptr(col), _plan_col32(plan, col),
                                         ctypes.c_void_p(rp), ctypes.c_void_p(rp + 8), ctypes.byref(ps), _ptr(y),
                                         y.stride(0) if n > 1 else max(k, y.stride(0)), _ptr(z),
                                         z.stride(0) if m > 1 else max(k, z.stride(0)), _ptr(z_arg), _ptr(workspace),
                                         0 if workspace is None else workspace.numel(),
                                         None if epilogue is None else ctypes.byref(epilogue), _stream(y.device))
    if check:
        _check(st, "fusedMM_csr_sweep_hip")
    return st


def spmm_sweep(rowptr, col, val, plan, y, reduce: str = "sum", workspace=None, row_scale=None, self_term=None, bias=None,
               relu=False):
    """Allocate outputs (+ workspace) and call the sweep boundary; returns (out, arg|None)."""
    rowptr = _dev(rowptr, "rowptr", torch.int64)
    col = _dev(col, "col", torch.int64)
    if val is not None:
        val = _dev(val, "val", torch.float32)
    y = y.contiguous()
    m, k = rowptr.numel() - 1, y.size(1)
    out = torch.empty((m, k), dtype=torch.float32, device=y.device)
    arg = torch.empty((m, k), dtype=torch.int64, device=y.device) if reduce in ("max", "min") else None
    if workspace is None:
        workspace = plan.workspace(reduce, k)
    ep = None
    if row_scale is not None or self_term is not None or bias is not None or relu:
        ep = Epilogue(None if row_scale is None else row_scale.data_ptr(), None if self_term is None else self_term.data_ptr(),
                      k if self_term is None else self_term.stride(0), None if bias is None else bias.data_ptr(), int(bool(relu)))
    fusedMM_csr_sweep_hip(MESSAGE[reduce], rowptr, col, val, plan, y, out, arg, workspace, ep)
    return out, arg


def fusedMM_csr_stream_hip(imessage: int, rowptr, nnz: int, plan, y, z, workspace=None, epilogue=None, check: bool = True) -> int:
    """Raw boundary call of the stream-form SpMM (sum / mean); ``plan`` is an isplib_amd.plan.StreamPlan."""
    assert y.is_cuda and y.dtype == torch.float32 and y.dim() == 2 and y.stride(1) == 1
    m, n, k = rowptr.numel() - 1, y.size(0), y.size(1)
    rp = rowptr.data_ptr()
    ps = plan.struct()
    with torch.cuda.device(y.device):
        st = lib().fusedMM_csr_stream_hip(int(imessage), m, n, k, int(nnz), ctypes.c_void_p(rp), ctypes.c_void_p(rp + 8),
                                          ctypes.byref(ps), _ptr(y), y.stride(0) if n > 1 else max(k, y.stride(0)), _ptr(z),
                                          z.stride(0) if m > 1 else max(k, z.stride(0)), _ptr(workspace),
                                          0 if workspace is None else workspace.numel(),
                                          None if epilogue is None else ctypes.byref(epilogue), _stream(y.device))
    if check:
        _check(st, "fusedMM_csr_stream_hip")
    return st


def spmm_stream(rowptr, nnz: int, plan, y, reduce: str = "sum", workspace=None, row_scale=None, self_term=None, bias=None,
                relu=False):
    """Allocate the output (+ workspace) and call the stream boundary; returns out."""
    rowptr = _dev(rowptr, "rowptr", torch.int64)
    y = y.contiguous()
    m, k = rowptr.numel() - 1, y.size(1)
    out = torch.empty((m, k), dtype=torch.float32, device=y.device)
    if workspace is None:
        workspace = plan.workspace()
    ep = None
    if row_scale is not None or self_term is not None or bias is not None or relu:
        ep = Epilogue(None if row_scale is None else row_scale.data_ptr(), None if self_term is None else self_term.data_ptr(),
                      k if self_term is None else self_term.stride(0), None if bias is None else bias.data_ptr(), int(bool(relu)))
    fusedMM_csr_stream_hip(MESSAGE[reduce], rowptr, nnz, plan, y, out, workspace, ep)
    return out


def fusedMM_csr_stream_minmax_hip(imessage: int, rowptr, nnz: int, plan, y, z, z_arg=None, workspace=None, check: bool = True) -> int:
    """Raw boundary call of the stream-form SpMM for max / min; ``plan`` must be built for stream_minmax_geometry()."""
    assert y.is_cuda and y.dtype == torch.float32 and y.dim() == 2 and y.stride(1) == 1
    m, n, k = rowptr.numel() - 1, y.size(0), y.size(1)
    rp = rowptr.data_ptr()
    ps = plan.struct()
    with torch.cuda.device(y.device):
        st = lib().fusedMM_csr_stream_minmax_hip(int(imessage), m, n, k, int(nnz), ctypes.c_void_p(rp), ctypes.c_void_p(rp + 8),
                                                 ctypes.byref(ps), _ptr(y), y.stride(0) if n > 1 else max(k, y.stride(0)), _ptr(z),
                                                 z.stride(0) if m > 1 else max(k, z.stride(0)), _ptr(z_arg), _ptr(workspace),
                                                 0 if workspace is None else workspace.numel(), _stream(y.device))
    if check:
        _check(st, "fusedMM_csr_stream_minmax_hip")
    return st


def spmm_stream_minmax(rowptr, nnz: int, plan, y, reduce: str = "max", workspace=None, want_arg: bool = True):
    """Allocate the outputs (+ workspace) and call the max / min stream boundary; returns (out, arg)."""
    rowptr = _dev(rowptr, "rowptr", torch.int64)
    y = y.contiguous()
    m, k = rowptr.numel() - 1, y.size(1)
    out = torch.empty((m, k), dtype=torch.float32, device=y.device)
    arg = torch.empty((m, k), dtype=torch.int64, device=y.device) if want_arg else None
    if workspace is None:
        workspace = plan.workspace(minmax=True)
    fusedMM_csr_stream_minmax_hip(MESSAGE[reduce], rowptr, nnz, plan, y, out, arg, workspace)
    return out, arg


def suggest_stream_minmax(m: int, n: int, nnz: int, k: int):
    """(streams, slices, chunk) when the stream schedule is expected to win for max / min on this shape, else None."""
    st, sl, ch = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    if not lib().isplib_suggest_stream_minmax(int(m), int(n), int(nnz), int(k), ctypes.byref(st), ctypes.byref(sl), ctypes.byref(ch)):
        return None
    return st.value, sl.value, ch.value


def stream_minmax_geometry(streams: int = 4):
    """(rows per wave, resident waves) of the max / min stream kernel for `streams` = 4 (64-column slots) or 8 (32-column)."""
    rpw, res = ctypes.c_int(0), ctypes.c_int(0)
    _check(lib().isplib_spmm_stream_minmax_geometry(int(streams), ctypes.byref(rpw), ctypes.byref(res)), "isplib_spmm_stream_minmax_geometry")
    return rpw.value, res.value


def hybrid_geometry(streams: int = 4):
    """(rows per wave, resident waves, table rows, hot-step cap per wave and slice) of the hybrid kernel (isplib_spmm_hybrid_geometry)."""
    v = [ctypes.c_int(0) for _ in range(4)]
    _check(exp_lib().isplib_spmm_hybrid_geometry(int(streams), *[ctypes.byref(x) for x in v]), "isplib_spmm_hybrid_geometry")
    return tuple(x.value for x in v)


def fusedMM_csr_hybrid_hip(imessage: int, rowptr, nnz: int, plan, y, z, workspace=None, epilogue=None, check: bool = True) -> int:
    """Raw boundary call of the hybrid-form SpMM (sum / mean, unit weights); ``plan`` is an isplib_amd.plan.HybridPlan."""
    assert y.is_cuda and y.dtype == torch.float32 and y.dim() == 2 and y.stride(1) == 1
    m, n, k = rowptr.numel() - 1, y.size(0), y.size(1)
    rp = rowptr.data_ptr()
    ps = plan.struct()
    with torch.cuda.device(y.device):
        st = exp_lib().fusedMM_csr_hybrid_hip(int(imessage), m, n, k, int(nnz), ctypes.c_void_p(rp), ctypes.c_void_p(rp + 8),
                                          ctypes.byref(ps), _ptr(y), y.stride(0) if n > 1 else max(k, y.stride(0)), _ptr(z),
                                          z.stride(0) if m > 1 else max(k, z.stride(0)), _ptr(workspace),
                                          0 if workspace is None else workspace.numel(),
                                          None if epilogue is None else ctypes.byref(epilogue), _stream(y.device))
    if check:
        _check(st, "fusedMM_csr_hybrid_hip")
    return st


def spmm_hybrid(rowptr, nnz: int, plan, y, reduce: str = "sum", workspace=None, row_scale=None, self_term=None, bias=None, relu=False):
    """Allocate the output (+ workspace) and call the hybrid boundary; returns out."""
    rowptr = _dev(rowptr, "rowptr", torch.int64)
    y = y.contiguous()
    m, k = rowptr.numel() - 1, y.size(1)
    out = torch.empty((m, k), dtype=torch.float32, device=y.device)
    if workspace is None:
        workspace = plan.workspace()
    ep = None
    if row_scale is not None or self_term is not None or bias is not None or relu:
        ep = Epilogue(None if row_scale is None else row_scale.data_ptr(), None if self_term is None else self_term.data_ptr(),
                      k if self_term is None else self_term.stride(0), None if bias is None else bias.data_ptr(), int(bool(relu)))
    fusedMM_csr_hybrid_hip(MESSAGE[reduce], rowptr, nnz, plan, y, out, workspace, ep)
    return out


def suggest_stream(m: int, n: int, nnz: int, k: int, weighted: bool = False):
    """(streams, slices, chunk) when the stream schedule is expected to win for this shape, else None
    (isplib_suggest_stream_weighted; `weighted`: the plan will carry edge weights)."""
    st, sl, ch = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    if not lib().isplib_suggest_stream_weighted(int(m), int(n), int(nnz), int(k), int(bool(weighted)), ctypes.byref(st), ctypes.byref(sl), ctypes.byref(ch)):
        return None
    return st.value, sl.value, ch.value


class _DevView:
    """A device array the C library owns, seen through __cuda_array_interface__ (no copy).  `owner` is the object whose
    destructor frees the array: torch.as_tensor keeps THIS object alive for as long as the tensor's storage lives (its
    deleter holds the reference), so holding the owner here ties the library's memory to every tensor that looks at it
    -- including the copies autograd saves for a backward that runs after the graph object is gone."""

    def __init__(self, ptr: int, count: int, typestr: str, owner=None):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (ptr, False), "version": 2}
        self._owner = owner


class NativeStreamPlan:
    """A stream plan built by the C library (isplib_stream_plan_build_hip) -- the torch-free host's counterpart of
    isplib_amd.plan.build_stream_plan; same interface as plan.StreamPlan for the boundary wrappers."""

    def __init__(self, rowptr, col, val, ncols: int, streams: int, slices: int, chunk: int, waves_per_gen: int = 0,
                 minmax: bool = False, fusedmm: bool = False):
        self._s = StreamPlanStruct()
        self.device = col.device
        m = rowptr.numel() - 1
        with torch.cuda.device(col.device):
            if fusedmm:       # the generic FusedMM kernel's geometry (isplib_fusedmm_stream_geometry); no weights in the plan
                _check(lib().isplib_stream_plan_build_fusedmm_hip(m, int(ncols), col.numel(), _ptr(rowptr), _ptr(col), int(streams), int(slices),
                                                                  int(chunk), int(waves_per_gen), ctypes.byref(self._s), _stream(col.device)),
                       "isplib_stream_plan_build_fusedmm_hip")
            elif minmax:
                _check(lib().isplib_stream_plan_build_minmax_hip(m, int(ncols), col.numel(), _ptr(rowptr), _ptr(col), _ptr(val),
                                                                 int(streams), int(slices), int(chunk), int(waves_per_gen), ctypes.byref(self._s),
                                                                 _stream(col.device)), "isplib_stream_plan_build_minmax_hip")
            else:
                _check(lib().isplib_stream_plan_build_hip(m, int(ncols), col.numel(), _ptr(rowptr), _ptr(col), _ptr(val), int(streams),
                                                          int(slices), int(chunk), int(waves_per_gen), ctypes.byref(self._s),
                                                          _stream(col.device)), "isplib_stream_plan_build_hip")
        for name in ("rows", "cols", "slices", "gens", "waves_per_gen", "rows_per_wave", "streams", "n_steps", "n_parts", "n_hub"):
            setattr(self, name, int(getattr(self._s, name)))

    def struct(self):
        return self._s

    def workspace(self, minmax: bool = False):
        L = lib()
        nbytes = (L.isplib_spmm_stream_minmax_workspace_bytes if minmax else L.isplib_spmm_stream_workspace_bytes)(ctypes.byref(self._s))
        return torch.empty(nbytes, dtype=torch.uint8, device=self.device)

    def set_values(self, val):
        with torch.cuda.device(self.device):
            _check(lib().isplib_stream_plan_set_values_hip(ctypes.byref(self._s), _ptr(val), _stream(self.device)), "isplib_stream_plan_set_values_hip")

    def array(self, name: str) -> torch.Tensor:
        """A copy of one of the plan's device arrays as a tensor."""
        nw = self.gens * self.waves_per_gen
        count, typestr = {"words": (self.n_steps * self.streams, "<i4"), "perm": (self.n_steps * self.streams, "<i4"),
                          "vals": (self.n_steps * self.streams, "<f4"), "wave_step_off": (nw + 1, "<i8"),
                          "wave_row": (nw * self.rows_per_wave, "<i4"), "wave_part": (nw * self.rows_per_wave, "<i4"),
                          "hub_row": (self.n_hub, "<i4"), "hub_off": (self.n_hub + 1, "<i4")}[name]
        ptr = getattr(self._s, name)
        if not ptr or count == 0:
            return torch.empty(0, device=self.device)
        return torch.as_tensor(_DevView(ptr, count, typestr), device=self.device).clone()

    def view(self, name: str, dtype: torch.dtype) -> torch.Tensor:
        """One of the plan's device arrays as a tensor WITHOUT a copy.  The memory belongs to this plan object; the tensor
        holds a reference to it (through its _DevView), so the arrays are freed only when the last tensor looking at them
        is gone -- never under a backward pass that still has them saved.  (close() is for owners that hand out no views.)"""
        nw = self.gens * self.waves_per_gen
        count, typestr = {"words": (self.n_steps * self.streams, "<i4"), "perm": (self.n_steps * self.streams, "<i4"),
                          "wave_step_off": (nw + 1, "<i8"), "wave_row": (nw * self.rows_per_wave, "<i4"),
                          "wave_part": (nw * self.rows_per_wave, "<i4"), "hub_row": (self.n_hub, "<i4"),
                          "hub_off": (self.n_hub + 1, "<i4")}[name]
        ptr = getattr(self._s, name)
        if not ptr or count == 0:
            return torch.empty(0, dtype=dtype, device=self.device)
        t = torch.as_tensor(_DevView(ptr, count, typestr, owner=self), device=self.device)
        assert t.dtype == dtype and t.data_ptr() == ptr, "zero-copy view of a library array"
        return t

    def close(self):
        if self._s.words or self._s.wave_row:
            lib().isplib_stream_plan_free(ctypes.byref(self._s))

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


def stream_geometry(streams: int = 4):
    """(rows per wave, resident waves) of the stream kernel for `streams` slots per wave (isplib_spmm_stream_geometry)."""
    rpw, res = ctypes.c_int(0), ctypes.c_int(0)
    _check(lib().isplib_spmm_stream_geometry(int(streams), ctypes.byref(rpw), ctypes.byref(res)), "isplib_spmm_stream_geometry")
    return rpw.value, res.value


def sweep_resident_waves(reduce: str, k: int, rows_per_wave: int = 16) -> int:
    return int(exp_lib().isplib_spmm_sweep_resident_waves(MESSAGE[reduce], int(k), int(rows_per_wave)))


class GraphHandle:
    """ctypes view of `isplib_graph` (include/isplib_hip.h): the torch-free host's per-graph object.  torch only
    supplies the device buffers here; plans, packed ids, CSC operands and workspace live inside the library."""

    def __init__(self, rowptr, col, val, ncols: int):
        self.rowptr = _dev(rowptr, "rowptr", torch.int64)       # borrowed by the handle: keep them alive
        self.col = _dev(col, "col", torch.int64)
        self.val = None if val is None else _dev(val, "val", torch.float32)
        self.m, self.n = self.rowptr.numel() - 1, int(ncols)
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.col.device):
            _check(lib().isplib_graph_create(self.m, self.n, self.col.numel(), _ptr(self.rowptr), _ptr(self.col),
                                             _ptr(self.val), ctypes.byref(self._h)), "isplib_graph_create")

    def set_slices(self, slices: int) -> None:
        _check(lib().isplib_graph_set_slices(self._h, int(slices)), "isplib_graph_set_slices")

    def set_row_order(self, order, order_t=None) -> None:
        """The plain kernel's row order of A / of A^T (int32, position -> row; None = index order, and no search for one)."""
        self.order = None if order is None else _dev(order, "order", torch.int32)           # borrowed by the handle: keep them alive
        self.order_t = None if order_t is None else _dev(order_t, "order_t", torch.int32)
        assert self.order is None or self.order.numel() == self.m
        assert self.order_t is None or self.order_t.numel() == self.n
        _check(lib().isplib_graph_set_row_order(self._h, _ptr(self.order), _ptr(self.order_t)), "isplib_graph_set_row_order")

    def set_values(self, val) -> None:
        """New weights for the same structure (another array, the same one edited in place, or None = unit weights)."""
        self.val = None if val is None else _dev(val, "val", torch.float32)
        assert self.val is None or self.val.numel() == self.col.numel()
        _check(lib().isplib_graph_set_values(self._h, _ptr(self.val)), "isplib_graph_set_values")

    def spmm(self, y: torch.Tensor, reduce: str = "sum"):
        """y: [n, k] with unit column stride; a row stride > k (a column block of a wider matrix) is passed on as ldy."""
        if not (y.is_cuda and y.dtype == torch.float32 and y.dim() == 2 and y.stride(1) == 1 and y.stride(0) >= y.size(1)):
            y = _dev(y, "y", torch.float32)
        k = y.size(1)
        ldy = y.stride(0) if y.size(0) > 1 else max(k, y.stride(0))
        out = torch.empty((self.m, k), dtype=torch.float32, device=y.device)
        arg = torch.empty((self.m, k), dtype=torch.int64, device=y.device) if reduce in ("max", "min") else None
        with torch.cuda.device(y.device):
            _check(lib().isplib_graph_spmm(self._h, MESSAGE[reduce], k, _ptr(y), ldy, _ptr(out), k, _ptr(arg), _stream(y.device)),
                   "isplib_graph_spmm")
        return out, arg

    def spmm_backward(self, dy: torch.Tensor, mean: bool = False) -> torch.Tensor:
        dy = _dev(dy, "dy", torch.float32)
        k = dy.size(1)
        dx = torch.empty((self.n, k), dtype=torch.float32, device=dy.device)
        with torch.cuda.device(dy.device):
            _check(lib().isplib_graph_spmm_backward(self._h, int(bool(mean)), k, _ptr(dy), k, _ptr(dx), k, _stream(dy.device)),
                   "isplib_graph_spmm_backward")
        return dx

    def sddmm(self, y: torch.Tensor, g: torch.Tensor, mean: bool = False) -> torch.Tensor:
        y, g = _dev(y, "y", torch.float32), _dev(g, "g", torch.float32)
        k = y.size(1)
        dval = torch.empty(self.col.numel(), dtype=torch.float32, device=y.device)
        with torch.cuda.device(y.device):
            _check(lib().isplib_graph_sddmm(self._h, int(bool(mean)), k, _ptr(y), k, _ptr(g), k, _ptr(dval), _stream(y.device)),
                   "isplib_graph_sddmm")
        return dval

    def close(self) -> None:
        if self._h:
            lib().isplib_graph_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass
