#!/usr/bin/env python3
"""Probe (CPU, numpy): error of a 32-term fp32 dot product computed from split operands, against float64 --
  bf16 x 3 pieces, six products (what csrc/stc_x3_frag.h does)   vs   fp16 x 2 pieces, three products (hh + hl + lh),
on operands shaped like the node kernels' (activations O(1), weights O(0.1), gradients O(1e-8) with and without a power-of-two scale).
Piece products are exact in fp32 (8 x 8 and 11 x 11 bit significands); sums are taken in fp32 like the MFMA accumulator."""
import numpy as np

rng = np.random.default_rng(0)


def bf16(a):                       # round-to-nearest-even to bfloat16, returned as float32
    u = a.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split_bf16x3(a):
    h = bf16(a); r = (a - h).astype(np.float32); m = bf16(r); l = bf16((r - m).astype(np.float32))
    return h, m, l


def split_fp16x2(a):
    h = a.astype(np.float16).astype(np.float32); l = (a - h).astype(np.float16).astype(np.float32)
    return h, l


def dot32(terms):                  # fp32 accumulation of a list of (A piece, B piece), smallest terms first
    acc = np.zeros(terms[0][0].shape[:-1], np.float32)
    for A, B in terms:
        for k in range(A.shape[-1]):
            acc = (acc + (A[..., k] * B[..., k]).astype(np.float32)).astype(np.float32)
    return acc


def run(name, A, B, scale_a=1.0):
    ref = (A.astype(np.float64) * B.astype(np.float64)).sum(-1)
    As = (A * np.float32(scale_a)).astype(np.float32)
    ah, am, al = split_bf16x3(A); bh, bm, bl = split_bf16x3(B)
    six = dot32([(al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm), (ah, bh)])
    fh, fl = split_fp16x2(As); gh, gl = split_fp16x2(B)
    three = dot32([(fl, gh), (fh, gl), (fh, gh)]) / np.float32(scale_a)
    plain = dot32([(A, B)])
    n = np.abs(ref).max()
    print(f'{name:58s} fp32 fma chain {np.abs(plain - ref).max() / n:.2e}   bf16x3, 6 products {np.abs(six - ref).max() / n:.2e}   '
          f'fp16x2, 3 products {np.abs(three - ref).max() / n:.2e}')


M = 4096
act = rng.uniform(-1, 1, (M, 32)).astype(np.float32)
w = (rng.standard_normal((M, 32)) * 0.1).astype(np.float32)
grad = (rng.standard_normal((M, 32)) * 1e-8).astype(np.float32)
small = (rng.uniform(-1, 1, (M, 32)) * 10.0 ** rng.uniform(-6, 0, (M, 32))).astype(np.float32)
run('activations x weights', act, w)
run('wide-range activations (1e-6 .. 1) x weights', small, w)
run('gradients 1e-8 x weights, unscaled', grad, w)
run('gradients 1e-8 x weights, scaled by 2^24', grad, w, scale_a=2.0 ** 24)
run('gradients 1e-8 x activations, scaled by 2^24', grad, act, scale_a=2.0 ** 24)
