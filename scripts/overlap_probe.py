#!/usr/bin/env python3
"""Would a layer's data gradient and filter gradient, which do not depend on each other, finish sooner side by side on two streams?  conv4_x /
conv5_x launch 212 resp. 252 workgroups on 256 CUs (one each): 17 % of the CUs idle in the first.  The pair back to back on one stream
against the pair on two streams.  The filter gradient on the second stream gets a workspace of its own (round 5 let both launches share the
library's: the stream-K data gradient's arrival words and the split-M slabs raced on one buffer -- advisor finding), the side-by-side results
are checked against the back-to-back ones, and the device status word is polled at the end.
usage: overlap_probe.py [layer]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from em_adapt_tensorflow_amd import ops
layer = sys.argv[1] if len(sys.argv) > 1 else "conv4_2"
H, Cin, Cout, k, rate = {"conv4_2": (41, 512, 512, 3, 1), "conv5_2": (41, 512, 512, 3, 2), "conv3_2": (81, 256, 256, 3, 1), "fc7": (41, 4096, 4096, 1, 1)}[layer]
B = 16
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn((B, H, H, Cin), device="cuda", generator=g).relu_().bfloat16()
dy = torch.randn((B, H, H, Cout), device="cuda", generator=g).bfloat16()
w = torch.randn((k, k, Cin, Cout), device="cuda", generator=g) * 0.05
wtd = ops.filter_tiles_bf16(w, dgrad=True)
dx = torch.empty_like(x)
dw = torch.empty_like(w)
db = torch.empty((Cout,), device="cuda")
def dgrad(): ops.conv_dgrad_bf16t(dy, wtd, (k, k), Cin, rate, gate=x, gate_scale=1.0, out=dx)
def wgrad(): ops.conv_wgrad_bf16(x, dy, k, k, rate, out=dw, bias_out=db)
ws_main, ws_side = ops._WS, ops.Workspace()   # (ops._WS is the pool every wrapper draws its workspace from)
def wgrad_side():
    ops._WS = ws_side
    try:
        wgrad()
    finally:
        ops._WS = ws_main
s2 = torch.cuda.Stream()
def seq():
    dgrad(); wgrad()
def par():
    e = torch.cuda.Event(); e.record()
    with torch.cuda.stream(s2):
        s2.wait_event(e)
        wgrad_side()
        e2 = torch.cuda.Event(); e2.record()
    dgrad()
    torch.cuda.current_stream().wait_event(e2)
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
seq(); torch.cuda.synchronize()
want = (dx.clone(), dw.clone(), db.clone())
par(); torch.cuda.synchronize()
assert torch.equal(dx, want[0]) and torch.equal(dw, want[1]) and torch.equal(db, want[2]), "side-by-side results differ"
for rep in range(3):
    print("%s  dgrad %.4f ms  wgrad %.4f ms  back to back %.4f ms  side by side %.4f ms" % (layer, timed(dgrad), timed(wgrad), timed(seq), timed(par)), flush=True)
assert ops.status_poll(clear=True, raise_on_error=False) == 0, "a launch recorded a failure in the device status word"
print("results identical side by side / back to back; status word clean")
