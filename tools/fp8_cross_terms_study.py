"""Numerical study (CPU, numpy): what would the split-fp16 layers lose if the two CROSS terms (hi x lo, lo x hi) ran on the fp8 matrix path
(e4m3 operands, one power-of-two scale per 32-element block as the MX-scaled MFMA applies), the hi x hi term staying on the f16 path?
Compares, for e = W2 relu(h) with K = 576 (the layer that is 75 % of the fused kernel's matrix work) and a five-layer chain of that shape:
  (a) fp64 reference; (b) the product arithmetic (fp16 hi / lo halves, three products, fp32 accumulate); (c) fp8 cross terms.
Errors are |x - ref| / max(1, |ref|) as in the parity tests, and relative to the layer's largest output.  profiles/round4_fused_experiments.md section 10."""
import numpy as np


def split16(x):
    hi = x.astype(np.float16)                       # round to nearest (the kernel truncates hi; same size of lo)
    lo = (x - hi.astype(np.float64)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def e4m3(x):
    """round to the nearest e4m3 value (3 mantissa bits, exponents 2^-6 .. 2^8, subnormals down to 2^-9, saturating at 448)"""
    s, a = np.sign(x), np.abs(x)
    e = np.floor(np.log2(np.maximum(a, 1e-300)))
    e = np.clip(e, -6, 8)
    q = 2.0 ** (e - 3)
    r = np.round(a / q) * q
    return s * np.minimum(r, 448.0)


def block_scaled_fp8(x, axis):
    """per 32-element block along `axis`: a power of two that brings the block's largest magnitude into [128, 256), then e4m3"""
    x = np.moveaxis(x, axis, -1)
    shp = x.shape
    xb = x.reshape(*shp[:-1], shp[-1] // 32, 32)
    m = np.abs(xb).max(axis=-1, keepdims=True)
    sc = 2.0 ** (7 - np.floor(np.log2(np.maximum(m, 1e-300))))
    y = e4m3(xb * sc) / sc
    return np.moveaxis(y.reshape(shp), -1, axis)


def layer(W, x, mode):
    """y[s, o] = sum_k W[o, k] x[s, k]; x, W already in fp16's window"""
    if mode == "ref":
        return x @ W.T
    wh, wl = split16(W)
    xh, xl = split16(x)
    acc = (xh @ wh.T).astype(np.float32).astype(np.float64)
    if mode == "f16x3":
        return acc + xh @ wl.T + xl @ wh.T
    cross = block_scaled_fp8(xh, 1) @ block_scaled_fp8(wl, 1).T + block_scaled_fp8(xl, 1) @ block_scaled_fp8(wh, 1).T
    return acc + cross


def report(name, y, ref):
    err = np.abs(y - ref)
    rel = err / np.maximum(1.0, np.abs(ref))
    print(f"   {name:34s} max |err| / max(1, |ref|) {rel.max():.2e}   max |err| / max |ref| {err.max() / np.abs(ref).max():.2e}   rms |err| / rms |ref| "
          f"{np.sqrt((err ** 2).mean()) / np.sqrt((ref ** 2).mean()):.2e}")


def main():
    rng = np.random.default_rng(0)
    S, K = 4096, 576
    print(f"one layer, K = {K}, {S} samples (inputs = relu of a unit normal, weights normal / sqrt(K)):")
    x = np.maximum(rng.standard_normal((S, K)), 0.0)
    W = rng.standard_normal((288, K)) / np.sqrt(K)
    ref = layer(W, x, "ref")
    report("fp16 hi/lo x3 (product)", layer(W, x, "f16x3"), ref)
    report("hi x hi on f16, cross terms on fp8", layer(W, x, "fp8"), ref)
    print("chain of five such layers (relu between, 288 -> 288), error of the last output:")
    Ws = [rng.standard_normal((288, 288)) * np.sqrt(2.0 / 288) for _ in range(5)]
    x0 = np.maximum(rng.standard_normal((S, 288)), 0.0)
    outs = {}
    for mode in ("ref", "f16x3", "fp8"):
        v = x0
        for i, Wl in enumerate(Ws):
            v = layer(Wl, v, mode)
            if i < 4:
                v = np.maximum(v, 0.0)
        outs[mode] = v
    report("fp16 hi/lo x3 (product)", outs["f16x3"], outs["ref"])
    report("hi x hi on f16, cross terms on fp8", outs["fp8"], outs["ref"])


if __name__ == "__main__":
    main()
