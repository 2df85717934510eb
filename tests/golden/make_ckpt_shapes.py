"""Generate tests/golden/ckpt_shapes.json from the checkpoint index shipped with
the reference (ckpts/GraphUNet/GraphUNet_03-10-2021__16-04-37/ckpt-1.index).

The index is a leveldb table of BundleEntryProto records; only names, dtypes
and shapes are extracted (the data shard is not shipped).  Nothing from the
file is executed.  Run here, where /root/reference exists:

    python tests/golden/make_ckpt_shapes.py
"""
import json
import os
import sys

INDEX = ('/root/reference/ckpts/GraphUNet/GraphUNet_03-10-2021__16-04-37/ckpt-1.index')
SUFFIX = '/.ATTRIBUTES/VARIABLE_VALUE'


def varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def block_entries(buf, offset, size):
    block = buf[offset:offset + size]
    n_restarts = int.from_bytes(block[-4:], 'little')
    end = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b''
    while pos < end:
        shared, pos = varint(block, pos)
        non_shared, pos = varint(block, pos)
        vlen, pos = varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def proto_fields(buf):
    pos = 0
    while pos < len(buf):
        tag, pos = varint(buf, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            val, pos = varint(buf, pos)
        elif wire == 2:
            ln, pos = varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wire == 5:
            val = buf[pos:pos + 4]
            pos += 4
        elif wire == 1:
            val = buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError(f'wire type {wire}')
        yield field, wire, val


def entry_shape(value):
    dtype, dims = None, []
    for field, wire, val in proto_fields(value):
        if field == 1 and wire == 0:
            dtype = val
        elif field == 2 and wire == 2:                 # TensorShapeProto
            for f2, w2, v2 in proto_fields(val):
                if f2 == 2 and w2 == 2:                # dim
                    size = 0
                    for f3, w3, v3 in proto_fields(v2):
                        if f3 == 1 and w3 == 0:
                            size = v3
                    dims.append(size)
    return dtype, dims


def main():
    with open(INDEX, 'rb') as f:
        buf = f.read()
    footer = buf[-48:]
    assert footer[-8:] == bytes.fromhex('57fb808b247547db'), 'not a leveldb table'
    pos = 0
    _, pos = varint(footer, pos)
    _, pos = varint(footer, pos)
    idx_off, pos = varint(footer, pos)
    idx_size, pos = varint(footer, pos)
    shapes = {}
    n_entries = 0
    for _, handle in block_entries(buf, idx_off, idx_size):
        off, p = varint(handle, 0)
        size, p = varint(handle, p)
        for key, value in block_entries(buf, off, size):
            n_entries += 1
            name = key.decode('utf-8', 'replace')
            if not name.endswith(SUFFIX) or '.OPTIMIZER_SLOT' in name:
                continue
            dtype, dims = entry_shape(value)
            if dtype != 1 or not name.startswith('net/'):   # float32 model variables only
                continue
            shapes[name[len('net/'):-len(SUFFIX)]] = dims
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ckpt_shapes.json')
    with open(out, 'w') as f:
        json.dump({'source': 'ckpt-1.index of GraphUNet_03-10-2021__16-04-37',
                   'n_index_entries': n_entries, 'shapes': shapes}, f, indent=1, sort_keys=True)
    n_params = 0
    for dims in shapes.values():
        k = 1
        for d in dims:
            k *= d
        n_params += k
    print(f'{n_entries} index entries, {len(shapes)} float32 model variables, {n_params} parameters -> {out}')


if __name__ == '__main__':
    sys.exit(main())
