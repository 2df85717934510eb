"""The C-ABI library loads and exports every symbol include/gunet_hip.h declares (no GPU needed),
and the host-side mirror keeps the reference's names."""
import ctypes
import os
import re

from conftest import ROOT


def _declared_symbols():
    with open(os.path.join(ROOT, 'include', 'gunet_hip.h')) as f:
        text = re.sub(r'/\*.*?\*/', '', f.read(), flags=re.S)
    return sorted(set(re.findall(r'\b(gunet_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_declared_symbols():
    import gunet_amd
    lib = ctypes.CDLL(gunet_amd.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 14
    for name in names:
        assert hasattr(lib, name), name
    assert set(names) == set(gunet_amd._lib.SIGNATURES), 'ctypes table and header disagree'
    assert gunet_amd._lib.lib.gunet_version() >= 1
    assert gunet_amd._lib.last_error() == ''


def test_header_is_plain_c(tmp_path):
    """include/gunet_hip.h compiles as C99 on its own, a C caller links against the library, and the
    structs have the layout the ctypes table assumes."""
    import subprocess
    import gunet_amd
    from gunet_amd import _lib
    src = tmp_path / 'use_header.c'
    src.write_text('''
#include "gunet_hip.h"
#include <stdio.h>
int main(void) {
  gunet_unet_weights w;
  gunet_quadgraph g;
  printf("%zu %zu %zu %d\\n", sizeof(g), sizeof(w.down[0]), sizeof(w), gunet_version());
  return gunet_plan_depth((const gunet_plan*)0) == GUNET_ERR_INVALID ? 0 : 1;
}
''')
    exe = tmp_path / 'use_header'
    libdir = os.path.dirname(gunet_amd.LIB_PATH)
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-pedantic', '-I', os.path.join(ROOT, 'include'), str(src),
                    '-o', str(exe), '-L', libdir, '-l:libgunet_hip.so', f'-Wl,-rpath,{libdir}',
                    '-Wl,--allow-shlib-undefined'], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert [int(v) for v in out[:3]] == [ctypes.sizeof(_lib.QuadGraphStruct), ctypes.sizeof(_lib.GnWeightsStruct),
                                         ctypes.sizeof(_lib.UNetWeightsStruct)]
    assert int(out[3]) >= 1


def test_mirror_keeps_reference_names():
    import gunet_amd
    from gunet_amd.layers import hybrid_gnn, graph_ops, get, LAYERS
    from gunet_amd.data import data_utils
    from gunet_amd.models.graph_unet import GraphUNet
    for name in ('GeoQuadrantGCN', 'GraphDownsamplingBlock', 'GraphUpsamplingBlock', 'GraphUpsampling2x2',
                 'graph_pooling_2x2', 'remove_duplicate_edges'):
        assert hasattr(hybrid_gnn, name)
    assert hasattr(graph_ops, 'get_sender_receiver') and hasattr(data_utils, 'sample_quadrant_subgraphs')
    assert data_utils.QUADRANT_IDS[:4] == [[0, 1, 2], [2, 3, 4], [4, 5, 6], [6, 7, 0]]
    assert get('GEO_QUADRANT_GCN') is hybrid_gnn.GeoQuadrantGCN and 'geo_quadrant_gcn' in LAYERS
    model = GraphUNet(32, out_units=96, depth=3, activation='relu', layer_type='geo_quadrant_gcn',
                      use_bias=True, use_global=True)
    prefixes = [p for p, _ in model._gn_layers()]
    assert len(prefixes) == 14 and prefixes[0] == 'downsampling_layers/0/gcns/0'
    assert prefixes[-1] == 'upsampling_layers/2/gcns/1'
    assert not model.upsampling_layers[-1].gcns[-1].has_glob_transform      # last layer: identity global
    assert all(l.has_glob_transform for _, l in model._gn_layers()[:-1])


def test_constants_of_the_header_match_the_ctypes_table_and_the_oracle(oracle):
    """Every GUNET_ACT_* / GUNET_MATMUL_* / GUNET_KERNELS_* value in the header is the one the Python
    side passes, and every activation the mirror accepts has an oracle function to be checked against."""
    from gunet_amd import _lib
    from gunet_amd.layers import hybrid_gnn
    with open(os.path.join(ROOT, 'include', 'gunet_hip.h')) as f:
        defines = dict(re.findall(r'#define\s+(GUNET_[A-Z0-9_]+)\s+\(?(-?\d+)\)?', f.read()))
    acts = {k[len('GUNET_ACT_'):]: int(v) for k, v in defines.items() if k.startswith('GUNET_ACT_')}
    assert len(acts) == 12 and sorted(acts.values()) == list(range(12))
    for name, code in acts.items():
        assert getattr(_lib, 'ACT_' + name) == code, name
    for name, code in hybrid_gnn._ACTIVATIONS.items():
        assert name in oracle.ACTIVATIONS, name
        assert code in acts.values()
    assert {acts[k.upper()] for k in hybrid_gnn._ACTIVATIONS if k} == set(acts.values())
    assert int(defines['GUNET_MAX_DEPTH']) == _lib.MAX_DEPTH
    assert int(defines['GUNET_STATIC_SCRATCH_BYTES']) == _lib.STATIC_SCRATCH_BYTES if 'GUNET_STATIC_SCRATCH_BYTES' in defines else True
