"""miq_act_calib_quant_fwd_static (the entry point that allocates nothing: its partial statistics live in a
device array inside the library) against miq_act_calib_quant_fwd (caller workspace), both called through the
C ABI with ctypes.  Every output must be bit-identical, on two calls in a row of which the second is smaller
(the static array keeps the first call's contents), with and without cached statistics; slabs beyond the
array's capacity must be refused before anything is launched.
"""
import ctypes

import pytest
import torch

from miq import _lib as L
from miq import functional as F
from miq import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DT = {"f32": torch.float32, "f16": torch.float16}


def _call(static, x, bits, metric, layout, cmin=None, cmax=None):
    """One forward through the C ABI; returns every output tensor of the call."""
    R, C = x.shape
    code = L.dtype_code(x)
    out = torch.empty_like(x)
    codes = torch.empty((R, C), dtype=torch.uint8, device=x.device)
    vec = torch.empty((5, C), dtype=x.dtype, device=x.device)
    head = (L.ptr(x), L.ptr(cmin), L.ptr(cmax), L.i32(1 if metric == "ema_minmax" else 0), L.ptr(layout.starts),
            L.i32(layout.n_clusters), L.i64(layout.covered), L.i32(bits))
    outs = (L.ptr(out), L.ptr(codes), L.ptr(vec[0]), L.ptr(vec[1]), L.ptr(vec[2]), L.ptr(vec[3]), L.ptr(vec[4]))
    tail = (L.i64(R), L.i64(C), L.i32(code), L.stream_of(x))
    if static:
        chan = torch.empty((2, C), dtype=torch.float32, device=x.device)
        L.call("miq_act_calib_quant_fwd_static", *head, *outs, L.ptr(chan[0]), L.ptr(chan[1]), *tail)
        cmn, cmx = chan[0], chan[1]
    else:
        rows = F.colstat_chunks(R, C, x.dtype) + 1
        ws = torch.empty((2, rows, C), dtype=torch.float32, device=x.device)
        L.call("miq_act_calib_quant_fwd", *head, L.ptr(ws), *outs, *tail)
        cmn, cmx = ws[0, rows - 1], ws[1, rows - 1]
    return {"out": out, "codes": codes, "cmin": vec[0], "cmax": vec[1], "scale": vec[2], "zp": vec[3], "rzp": vec[4],
            "chan_min": cmn, "chan_max": cmx}


def _same(a, b):
    if a.is_floating_point():
        return torch.equal(torch.nan_to_num(a.float(), nan=12345.0), torch.nan_to_num(b.float(), nan=12345.0))
    return torch.equal(a, b)


@pytest.mark.parametrize("dt", ["f32", "f16"])
@pytest.mark.parametrize("metric", ["ema_minmax", "minmax"])
def test_static_scratch_matches_workspace_variant_twice_in_a_row(dt, metric):
    dtype = DT[dt]
    # a large slab first, then a smaller one with another cluster layout: the static array still holds the
    # first call's partial statistics
    seq = [((2048, 4096), [128] * 32, 4), ((333, 1000), [50] * 10 + [20] * 10 + [7, 93], 4), ((77, 520), [1] * 20 + [400], 6)]
    cached = {}
    for n, (shape, counts, bits) in enumerate(seq):
        x = synth.activation(shape, 700 + n, dtype).to(DEV)
        layout = F.ClusterLayout.get(counts, shape[1], DEV)
        for use_cache in (False, True):
            cm, cx = cached.get(shape, (None, None)) if use_cache else (None, None)
            a = _call(True, x, bits, metric, layout, cm, cx)
            b = _call(False, x, bits, metric, layout, cm, cx)
            torch.cuda.synchronize()
            for k in b:
                assert _same(a[k], b[k]), f"call {n} cache={use_cache}: {k} differs"
            cached[shape] = (b["cmin"].clone(), b["cmax"].clone())


def test_static_scratch_refuses_slabs_beyond_its_capacity():
    cap = int(L.lib().miq_act_static_capacity())
    assert cap == 2 << 20
    R, C = 64, 1 << 20  # needs 2 * (nchunk + 1) * C floats > capacity
    assert int(L.lib().miq_act_workspace(L.i64(R), L.i64(C), L.i32(L.F32))) > cap
    x = torch.zeros((R, C), dtype=torch.float32, device=DEV)
    layout = F.ClusterLayout.get(None, C, DEV)
    with pytest.raises(L.MiqError, match="static buffer"):
        _call(True, x, 4, "minmax", layout)
    # and the library still works afterwards
    y = synth.activation((64, 256), 9, torch.float32).to(DEV)
    lay = F.ClusterLayout.get([64] * 4, 256, DEV)
    a, b = _call(True, y, 4, "minmax", lay), _call(False, y, 4, "minmax", lay)
    assert _same(a["out"], b["out"]) and _same(a["scale"], b["scale"])


def test_static_entry_point_allocates_nothing_and_runs_under_stream_capture():
    """After miq_act_static_init() the call makes no device allocation: free device memory is unchanged across it, and
    it can be captured into a HIP graph (an allocation would end the capture with an error) whose replay gives the
    bits of the eager call."""
    with L.on_device(torch.device(DEV)):
        L.call("miq_act_static_init")
    shape, counts, bits = (512, 1024), [128] * 8, 4
    x = synth.activation(shape, 731, torch.float32).to(DEV)
    layout = F.ClusterLayout.get(counts, shape[1], DEV)
    ref = _call(False, x, bits, "minmax", layout)
    R, C = shape
    out = torch.empty_like(x)
    codes = torch.empty((R, C), dtype=torch.uint8, device=DEV)
    vec = torch.empty((5, C), dtype=x.dtype, device=DEV)
    chan = torch.empty((2, C), dtype=torch.float32, device=DEV)
    side = torch.cuda.Stream()

    def call():
        L.call("miq_act_calib_quant_fwd_static", L.ptr(x), L.ptr(None), L.ptr(None), L.i32(0), L.ptr(layout.starts),
               L.i32(layout.n_clusters), L.i64(layout.covered), L.i32(bits), L.ptr(out), L.ptr(codes), L.ptr(vec[0]),
               L.ptr(vec[1]), L.ptr(vec[2]), L.ptr(vec[3]), L.ptr(vec[4]), L.ptr(chan[0]), L.ptr(chan[1]), L.i64(R), L.i64(C),
               L.i32(L.dtype_code(x)), L.stream_of(x))

    with torch.cuda.stream(side):
        call()  # the stream's own first launch lets the runtime set up its queue
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    with torch.cuda.stream(side):
        call()
    side.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before
    assert _same(out, ref["out"]) and _same(vec[2], ref["scale"])
    out.zero_()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        call()
    g.replay()
    torch.cuda.synchronize()
    assert _same(out, ref["out"]) and _same(codes, ref["codes"]) and _same(vec[4], ref["rzp"])
    assert _same(chan[0], ref["chan_min"]) and _same(chan[1], ref["chan_max"])
