"""A/B of one denoising forward (CFG batch of 2 x 24 latent frames) over settings of mimo_amd.ops module knobs:
  python tools/ab_forward.py CHUNK_TOKENS=0 CHUNK_TOKENS=65536 [--size 512]
Each setting is timed three times, interleaved; prints the per-setting best and the gemm/attention family times."""
import os
import sys

if any(a.startswith("env:") or ",env:" in a for a in sys.argv[1:]):  # environment knobs exist in the tune build only
    os.environ.setdefault("MIMO_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mimo_amd", "libmimo_hip_tune.so"))
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mimo_amd import ops  # noqa: E402


def main():
    size = 512
    args = [a for a in sys.argv[1:]]
    if "--size" in args:
        i = args.index("--size")
        size = int(args[i + 1])
        del args[i:i + 2]
    settings = [dict(kv.split("=") for kv in a.split(",")) for a in args] or [{}]
    dev = torch.device("cuda:0")
    pipe = bench.build_pipeline(dev, torch.float16)
    best = [None] * len(settings)
    defaults = {k: getattr(ops, k) for st in settings for k in st if not k.startswith("env:")}
    env_keys = {k[4:] for st in settings for k in st if k.startswith("env:")}
    for rnd in range(3):
        for i, st in enumerate(settings):
            for k, v in defaults.items():  # every knob any setting touches starts from its default
                setattr(ops, k, v)
            for k in env_keys:
                os.environ.pop(k, None)
            for k, v in st.items():
                if k.startswith("env:"):  # e.g. env:MIMO_GEMM_8P=0 (read by the tune library at every launch)
                    os.environ[k[4:]] = v
                    continue
                cur = defaults[k]
                setattr(ops, k, (v not in ("0", "false", "False", "")) if isinstance(cur, bool) else type(cur)(v))
            t, fl, n, fam = bench.measure_forward(pipe, dev, torch.float16, size, iters=3)
            if best[i] is None or t < best[i][0]:
                best[i] = (t, n, fam)
    for st, (t, n, fam) in zip(settings, best):
        print(f"{st}: forward {t*1e3:.2f} ms, {n} launches; " +
              "; ".join(f"{k} {d['ms']:.2f} ms / {d['launches']}" for k, d in fam.items()), flush=True)


if __name__ == "__main__":
    main()
