#!/usr/bin/env python3
"""Statistics of the built-in dropout generator (csrc/igemm_common.h dropout_keep) through the fused forward epilogue:
keep rate, agreement between different seeds (should be keep^2 + (1-keep)^2), lag-1 correlation along the channel and
pixel axes, rate per channel.  usage: dropout_quality.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from em_adapt_tensorflow_amd import ops
B, H, W, C = 4, 41, 41, 1024
x = torch.ones(B, H, W, 16, device="cuda"); w = torch.ones(1, 1, 16, C, device="cuda"); b = torch.ones(C, device="cuda")
def mask(seed, keep=0.5):
    return (ops.conv_fwd(x, w, b, 1, relu=True, keep_prob=keep, drop_seed=seed) != 0).float()
m1, m2, m3 = mask(7), mask(8), mask((5 << 32) + 7)
print("keep rate %.5f (0.5)" % float(m1.mean()), " rate at keep 0.9: %.5f" % float(mask(3, 0.9).mean()))
print("agreement seed 7 vs 8: %.5f, 7 vs (5<<32)+7: %.5f (0.5 expected)" % (float((m1 == m2).float().mean()), float((m1 == m3).float().mean())))
f = m1.reshape(-1, C)
print("lag-1 correlation along channels %.5f, along pixels %.5f" % (
    float(((f[:, 1:] - .5) * (f[:, :-1] - .5)).mean() * 4), float(((f[1:] - .5) * (f[:-1] - .5)).mean() * 4)))
pc = f.mean(0)
print("per-channel rate: min %.4f max %.4f (n = %d per channel, sigma = %.4f)" % (float(pc.min()), float(pc.max()), f.shape[0], 0.5 / f.shape[0] ** 0.5))
# cross-seed structure: the masks of two seeds must not be shifted copies of one another (the generator once mixed seed and
# index linearly, which made every seed the same sequence at an offset).  Circular cross-correlation at every lag, by FFT.
def max_xcorr(a, b):
    a = a.reshape(-1).double() - a.double().mean(); b = b.reshape(-1).double() - b.double().mean()
    n = a.numel()
    c = torch.fft.irfft(torch.fft.rfft(a) * torch.fft.rfft(b).conj(), n=n) / (a.norm() * b.norm())
    return float(c.abs().max()), n
for sa, sb in ((7, 8), (64 * 3 + 14, 64 * 8 + 14), (7, (1 << 32) + 7)):
    c, n = max_xcorr(mask(sa), mask(sb))
    print("max |cross-correlation| over all %d lags, seeds %d / %d: %.5f (noise level ~%.5f)" % (n, sa, sb, c, (2 * torch.log(torch.tensor(float(n))) / n) ** 0.5))
c, n = max_xcorr(m1, m1)
print("self-correlation (sanity, must be 1): %.5f" % c)
