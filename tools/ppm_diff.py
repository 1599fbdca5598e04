#!/usr/bin/env python3
"""Compares two P3 PPM frames (the reference's output format, ImageGenerator.hs:100 + showPixel): per-channel 8-bit
differences, RMSE and PSNR.  Used by bench/run_reference.sh on a box with GHC to pin the oracle against the real
Haskell binary: deterministic configs (antialiasing off, maxBounces <= 1) must agree to <= 1 level on >= 99.99 % of the
values; stochastic configs (independent RNGs) are compared in the mean: PSNR >= --psnr dB.

    tools/ppm_diff.py a.ppm b.ppm [--deterministic] [--psnr 30]
"""
import argparse
import sys

import numpy as np


def read_p3(path):
    tok = open(path, "rb").read().split()
    if tok[0] != b"P3":
        raise ValueError(f"{path}: not a P3 file")
    w, h, mx = int(tok[1]), int(tok[2]), int(tok[3])
    a = np.array(tok[4:4 + 3 * w * h], dtype=np.int64)
    if a.size != 3 * w * h:
        raise ValueError(f"{path}: {a.size} values, expected {3 * w * h}")
    return a.reshape(h, w, 3), mx


def compare(a, b):
    d = np.abs(a - b)
    mse = float(np.mean((a - b).astype(np.float64) ** 2))
    return {"max_abs": int(d.max()), "frac_equal": float((d == 0).mean()), "frac_within_1": float((d <= 1).mean()),
            "rmse": mse ** 0.5, "psnr_db": float("inf") if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("a"); ap.add_argument("b")
    ap.add_argument("--deterministic", action="store_true")
    ap.add_argument("--psnr", type=float, default=30.0)
    args = ap.parse_args()
    (a, _), (b, _) = read_p3(args.a), read_p3(args.b)
    if a.shape != b.shape:
        print(f"shapes differ: {a.shape} vs {b.shape}"); return 2
    r = compare(a, b)
    print(" ".join(f"{k}={v:.6g}" for k, v in r.items()))
    ok = r["frac_within_1"] >= 0.9999 if args.deterministic else r["psnr_db"] >= args.psnr
    print("PASS" if ok else "FAIL")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
