"""ctypes binding of libgunet_hip.so (the C ABI declared in include/gunet_hip.h).

The library is the only compute path: if it is missing, importing this module
raises, and nothing falls back to PyTorch or to a CPU implementation.  PyTorch
is used by the callers for device memory and streams only.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GUNET_HIP_LIBRARY: another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get('GUNET_HIP_LIBRARY') or os.path.join(_HERE, 'libgunet_hip.so')

ACT_LINEAR = 0
ACT_RELU = 1
ACT_TANH, ACT_SIGMOID, ACT_ELU, ACT_SOFTPLUS, ACT_SELU, ACT_SWISH = 2, 3, 4, 5, 6, 7
ACT_GELU, ACT_SOFTSIGN, ACT_EXPONENTIAL, ACT_HARD_SIGMOID = 8, 9, 10, 11

MATMUL_MODES = {'f32': 0, 'split3': 1, 'split2': 2, 'bf16': 3}

i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
f32p = C.POINTER(C.c_float)


class GunetError(RuntimeError):
    """A libgunet_hip entry point returned a non-zero status."""

    def __init__(self, fn, code, message):
        super().__init__(f'{fn} failed with status {code}: {message}')
        self.code = code


class QuadGraphStruct(C.Structure):
    _fields_ = [('n_nodes', C.c_int32),
                ('n_edges', C.c_int32 * 4),
                ('edge_index', C.c_void_p * 4),
                ('rowptr', C.c_void_p * 4),
                ('perm', C.c_void_p * 4),
                ('in_table', C.c_void_p)]


class GnWeightsStruct(C.Structure):
    _fields_ = [('c_in', C.c_int32), ('f_e', C.c_int32), ('g', C.c_int32), ('units', C.c_int32),
                ('node_emb_w', C.c_void_p * 4), ('node_emb_b', C.c_void_p * 4),
                ('edge_w', C.c_void_p * 4), ('edge_b', C.c_void_p * 4),
                ('glob_w', C.c_void_p * 4), ('glob_b', C.c_void_p * 4),
                ('node_w', C.c_void_p), ('node_b', C.c_void_p),
                ('globt_w', C.c_void_p), ('globt_b', C.c_void_p),
                ('global_scale_nodes', C.c_float), ('global_scale_edges', C.c_float)]


MAX_DEPTH = 8
STATIC_SCRATCH_BYTES = 64 << 20      # GUNET_STATIC_SCRATCH_BYTES


class UNetWeightsStruct(C.Structure):
    _fields_ = [('depth', C.c_int32), ('c_in', C.c_int32), ('out_units', C.c_int32),
                ('in_kernel', C.c_void_p), ('in_bias', C.c_void_p),
                ('out_kernel', C.c_void_p), ('out_bias', C.c_void_p),
                ('down', GnWeightsStruct * MAX_DEPTH),
                ('bottleneck', GnWeightsStruct * 2),
                ('up_sampling', GnWeightsStruct * MAX_DEPTH),
                ('up_gcn', (GnWeightsStruct * 2) * MAX_DEPTH)]


VoidP4 = C.c_void_p * 4
I64x4 = C.c_int64 * 4
PROF_CLASSES = 14
F64xP = C.c_double * PROF_CLASSES
I64xP = C.c_int64 * PROF_CLASSES

# name -> (restype, argtypes); every symbol declared in include/gunet_hip.h
SIGNATURES = {
    'gunet_last_error': (C.c_char_p, []),
    'gunet_version': (C.c_int, []),
    'gunet_quadrant_partition': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(VoidP4),
                                           C.POINTER(I64x4), C.c_void_p]),
    'gunet_csr_build': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    'gunet_csr_from_keys': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    'gunet_in_edge_table': (C.c_int, [C.POINTER(QuadGraphStruct), C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    'gunet_cluster_ids': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                    C.POINTER(C.c_int64), C.c_void_p]),
    'gunet_pool_edges': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                   C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]),
    'gunet_upsample_edges': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                       C.POINTER(VoidP4), C.POINTER(I64x4), C.c_void_p]),
    'gunet_get_sender_receiver_f32': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                                                C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    'gunet_segment_max_f32': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    'gunet_segment_sum_f32': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    'gunet_release_temporaries': (C.c_int, []),
    'gunet_set_matmul_mode': (C.c_int, [C.c_int]),
    'gunet_get_matmul_mode': (C.c_int, []),
    'gunet_set_kernel_choice': (C.c_int, [C.c_int]),
    'gunet_dense_workspace_bytes': (C.c_size_t, [C.c_int32, C.c_int32]),
    'gunet_dense_forward_f32': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                          C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t,
                                          C.c_void_p]),
    'gunet_gn_layer_workspace_bytes': (C.c_size_t, [C.POINTER(QuadGraphStruct), C.c_int32,
                                                    C.c_int32]),
    'gunet_gn_layer_forward_f32': (C.c_int, [C.POINTER(QuadGraphStruct),
                                             C.POINTER(GnWeightsStruct), C.c_int32, C.c_void_p,
                                             C.c_void_p, C.c_int32, C.POINTER(VoidP4), C.c_int32, C.c_void_p,
                                             C.c_int32, C.c_void_p, C.POINTER(VoidP4), C.c_void_p,
                                             C.c_void_p, C.c_size_t, C.c_void_p]),
    'gunet_gn_upsample_forward_f32': (C.c_int, [C.POINTER(QuadGraphStruct), C.c_int32,
                                                C.POINTER(GnWeightsStruct), C.c_int32, C.c_void_p,
                                                C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_size_t, C.c_void_p]),
    'gunet_plan_create': (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(VoidP4), C.POINTER(I64x4), C.c_int32,
                                    C.POINTER(C.c_void_p), C.c_void_p]),
    'gunet_plan_destroy': (C.c_int, [C.c_void_p]),
    'gunet_plan_depth': (C.c_int, [C.c_void_p]),
    'gunet_plan_level': (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(QuadGraphStruct), C.POINTER(C.c_void_p)]),
    'gunet_plan_pooling': (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(VoidP4)]),
    'gunet_plan_upsampling': (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(QuadGraphStruct),
                                        C.POINTER(C.c_void_p)]),
    'gunet_plan_pool_features_f32': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                               C.POINTER(VoidP4), C.c_int32, C.c_void_p, C.POINTER(VoidP4),
                                               C.c_void_p]),
    'gunet_unet_workspace_bytes': (C.c_size_t, [C.c_void_p, C.POINTER(UNetWeightsStruct), C.c_int32]),
    'gunet_unet_forward_f32': (C.c_int, [C.c_void_p, C.POINTER(UNetWeightsStruct), C.c_int32, C.c_void_p,
                                         C.POINTER(VoidP4), C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                         C.POINTER(VoidP4), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'gunet_unet_forward_static_f32': (C.c_int, [C.c_void_p, C.POINTER(UNetWeightsStruct), C.c_int32, C.c_void_p,
                                                C.POINTER(VoidP4), C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                                C.POINTER(VoidP4), C.c_void_p, C.c_void_p]),
    'gunet_street_encoding_f32': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int64, C.c_void_p, C.c_void_p]),
    'gunet_edge_features_f32': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                          C.c_void_p]),
    'gunet_model_inputs_workspace_bytes': (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    'gunet_model_inputs_f32': (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                         C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_size_t, C.c_void_p]),
    'gunet_image_to_graph': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.POINTER(C.c_int64 * 2), C.c_void_p]),
    'gunet_prediction_image_u8': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                            C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p,
                                            C.c_void_p]),
    'gunet_error_sums_workspace_bytes': (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    'gunet_error_sums_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p,
                                       C.c_void_p, C.c_size_t, C.c_void_p]),
    'gunet_profile_enable': (C.c_int, [C.c_int]),
    'gunet_profile_collect': (C.c_int, [C.POINTER(F64xP), C.POINTER(I64xP), C.POINTER(F64xP)]),
    'gunet_profile_class_name': (C.c_char_p, [C.c_int]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f'{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, "csrc")}` '
            '(or python -c "import __graft_entry__ as g; g.build()"); there is no fallback path')
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


lib = _load()


def last_error():
    return lib.gunet_last_error().decode('utf-8', 'replace')


def check(fn_name, status):
    if status != 0:
        raise GunetError(fn_name, status, last_error())


def call(fn_name, *args):
    """Call a status-returning entry point and raise GunetError on failure."""
    check(fn_name, getattr(lib, fn_name)(*args))


def ptr(t):
    """Device address of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def ptr4(tensors):
    return VoidP4(*[(t.data_ptr() if t is not None else None) for t in tensors])


def current_stream(device):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def profile_enable(on):
    call('gunet_profile_enable', 1 if on else 0)


def profile_collect():
    """{class name: (milliseconds, launches, algorithmic work)} since the previous collect; work is
    FLOPs for the matrix classes and bytes for aggregate / segment_reduce."""
    ms, n, work = F64xP(), I64xP(), F64xP()
    call('gunet_profile_collect', C.byref(ms), C.byref(n), C.byref(work))
    return {lib.gunet_profile_class_name(c).decode(): (float(ms[c]), int(n[c]), float(work[c]))
            for c in range(PROF_CLASSES)}


def set_matmul_mode(mode):
    """'f32' (exact float32 MFMA), 'split3' (float32-class on bf16 matrix cores), 'split2', 'bf16'."""
    call('gunet_set_matmul_mode', MATMUL_MODES[mode] if isinstance(mode, str) else int(mode))


def get_matmul_mode():
    code = lib.gunet_get_matmul_mode()
    return {v: k for k, v in MATMUL_MODES.items()}[code]


KERNEL_CHOICES = {'auto': 0, 'resident': 1, 'tiled': 2}


def set_kernel_choice(choice):
    """'auto' (default), 'resident' (weights-resident / message-summing GEMMs wherever they apply) or
    'tiled' (never): a tuning and test knob, results agree to rounding."""
    call('gunet_set_kernel_choice', KERNEL_CHOICES[choice] if isinstance(choice, str) else int(choice))
