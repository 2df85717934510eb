"""A compiled caller of the C ABI (examples/unet_forward.cpp: HIP runtime + gunet_hip.h only, no
Python or torch in the process): quadrant partition, plan, weight table, one-call forward.  The
program dumps its graph, inputs, weights and prediction; the oracle recomputes the prediction
from the dump.  Tolerance (fp32): max|got - want| / max|want| <= 1e-5."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, rel_err

pytestmark = pytest.mark.gpu

EXE = os.path.join(ROOT, 'examples', 'unet_forward')
DTYPES = {0: np.float32, 1: np.int32, 2: np.int64}


def read_dump(path):
    out = {}
    with open(path, 'rb') as f:
        while True:
            head = f.read(4)
            if not head:
                return out
            name = f.read(struct.unpack('<I', head)[0]).decode()
            ndim = struct.unpack('<I', f.read(4))[0]
            dims = struct.unpack(f'<{ndim}q', f.read(8 * ndim))
            dtype = np.dtype(DTYPES[struct.unpack('<I', f.read(4))[0]])
            count = int(np.prod(dims, dtype=np.int64))
            out[name] = np.frombuffer(f.read(count * dtype.itemsize), dtype=dtype).reshape(dims)


@pytest.mark.parametrize('height,width,batch', [(37, 29, 2), (20, 64, 1)])
def test_compiled_caller_against_oracle(oracle, tmp_path, height, width, batch):
    assert os.path.exists(EXE), f'{EXE} is missing: make -C graph-unet-traffic-prediction_amd/csrc'
    path = str(tmp_path / 'dump.bin')
    run = subprocess.run([EXE, str(height), str(width), str(batch), path], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert 'runs identical: yes' in run.stdout
    d = read_dump(path)
    quads = oracle.sample_quadrant_subgraphs(d['edge_index'], d['edge_direction'])
    W = {k: v for k, v in d.items() if k.endswith('/kernel') or k.endswith('/bias')}
    assert set(W) == {k for k, shp in oracle.graph_unet_shapes().items() if len(shp) > 0}
    for b in range(batch):
        graph = {'edge_index': list(quads), 'edges': [d[f'edges/{q}'][b] for q in range(4)],
                 'node_loc': d['node_loc'], 'global': d['global'][b:b + 1]}
        want, _ = oracle.predict(d['x'][b], graph, W)
        assert rel_err(d['prediction'][b], want) <= 1e-5, b
