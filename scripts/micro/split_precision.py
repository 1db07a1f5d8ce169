"""Host-side emulation of the three-product split on a 128-term dot product of post-relu activations: 16-bit operands as bf16 (the kernel's split today) and
as fp16 (11-bit significands: same matrix-core rate on gfx950), the latter with and without its lo terms pre-scaled by 2^11 into a second accumulator.
    python scripts/micro/split_precision.py
Round 5 numbers (relative to sum |terms|): bf16x3 3e-6 .. 5e-6 at every scale; fp16x3 1.2e-7 at O(1) activations, 1.5e-6 at 1e-2 (the lo terms fall into
fp16's subnormals); fp16x3 with scaled lo terms 6e-8 .. 8e-8 everywhere -- float32 level.  What the emulation does not show: fp16 overflows at 65 504."""
import numpy as np
import torch


def bf16(x):
    return torch.from_numpy(x.astype(np.float32)).to(torch.bfloat16).to(torch.float32).numpy().astype(np.float64)


def fp16(x):
    return x.astype(np.float32).astype(np.float16).astype(np.float64)


def main():
    rng = np.random.default_rng(0)
    M, K = 4000, 128
    for scale in (1.0, 1e-2, 30.0):
        x = np.maximum((rng.standard_normal((M, K)) * scale).astype(np.float32).astype(np.float64), 0)
        w = (rng.standard_normal((K,)) * np.sqrt(2 / K)).astype(np.float32).astype(np.float64)
        exact = x @ w
        xh, wh = bf16(x), bf16(w)
        b3 = xh @ wh + xh @ bf16(w - wh) + bf16(x - xh) @ wh
        xh, wh = fp16(x), fp16(w)
        f3 = xh @ wh + xh @ fp16(w - wh) + fp16(x - xh) @ wh
        s = 2.0 ** 11
        f3s = xh @ wh + (xh @ fp16((w - wh) * s) + fp16((x - xh) * s) @ wh) / s
        den = np.abs(x) @ np.abs(w) + 1e-30
        print("activation scale %-6g max|x| %-8.3g relative error: bf16x3 %.2e   fp16x3 %.2e   fp16x3, lo terms scaled %.2e" % (
            scale, x.max(), np.max(np.abs(b3 - exact) / den), np.max(np.abs(f3 - exact) / den), np.max(np.abs(f3s - exact) / den)))


if __name__ == "__main__":
    main()
