"""Longer run of the generator of tests/test_fuzz_gpu.py: python tools/fuzz_executor.py [seeds] [max_steps]

FUZZ_CONSTANTS=1: every input is uploaded once and MARKED CONSTANT, and each tree is evaluated twice on
the device -- the second time through whatever the executor cached of it (element-wise values of
constants, sums of constants, wide operands of concatenated products and the products with them) --
both against the float64 oracle."""
import builtins
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_fuzz_gpu as F                                    # noqa: E402
from bayesic_amd.algebra.device_backend import DeviceBackend  # noqa: E402
from bayesic_amd.device import Context                        # noqa: E402
from oracle.einsum_eval import NumpyBackend                   # noqa: E402


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    max_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 14
    scale_dims = int(os.environ.get("FUZZ_SCALE", "1"))       # e.g. 64: extents that reach the 16-byte / MFMA paths
    if scale_dims > 1:
        for n in list(F.SHAPES):
            F.SHAPES[n] = tuple(d * scale_dims for d in F.SHAPES[n])
    dev = DeviceBackend(Context(0))
    constants = os.environ.get("FUZZ_CONSTANTS", "0") == "1"
    resident = {}
    if constants:
        import torch
        for name, value in F.inputs().items():
            arr = np.asarray(value)
            resident[name] = (dev.from_host(arr.astype(np.float32), "float32", arr.ndim), arr)
        for t, _ in resident.values():
            if isinstance(t, torch.Tensor):
                dev.mark_constant(t)
                dev.mark_constant_tensor(t)
    bad = ran = 0
    for seed in range(1000, 1000 + seeds):
        made = F.Grower(seed).grow(2 + seed % max_steps)
        if made is None:
            continue
        expr, _ = made
        vals = {n: v for n, v in F.inputs().items() if n in expr.input_types}
        try:
            want = np.asarray(expr.compile(NumpyBackend(np.float64))(**vals), np.float64)
            if not np.isfinite(want).all():
                continue
            if constants:
                f = expr.compile(dev).device_fn
                dvals = {n: resident[n][0] for n in vals}
                first = np.asarray(dev.to_host(f(**dvals)), np.float64)
                got = np.asarray(dev.to_host(f(**dvals)), np.float64)
                if first.shape != got.shape or not np.allclose(first, got, rtol=1e-5, atol=1e-6 * (1 + np.abs(want).max())):
                    bad += 1
                    print("CACHED != FIRST seed %d: %r" % (seed, expr), flush=True)
                if got.ndim == 0 and want.ndim > 0:
                    got = got.reshape((1,) * want.ndim)
                if got.shape != want.shape:
                    got = np.broadcast_to(got, want.shape)
            else:
                got = np.asarray(expr.compile(dev)(**vals), np.float64)
            ran += 1
            scale = builtins.max(float(np.abs(want).max()) if want.size else 0.0, 1e-3)
            err = float(np.abs(got - want).max()) if want.size else 0.0
            if got.shape != want.shape or not err <= 2e-4 * scale:
                bad += 1
                print("MISMATCH seed %d err %g scale %g: %r" % (seed, err, scale, expr), flush=True)
        except Exception as exc:   # noqa: BLE001
            bad += 1
            print("EXCEPTION seed %d %s: %s | %r" % (seed, type(exc).__name__, exc, expr), flush=True)
    print("ran %d trees, %d problems" % (ran, bad))


if __name__ == "__main__":
    main()
