#!/usr/bin/env python3
"""Dev-only A/B driver for ONE GPU box: variants = environment settings and / or compile-time flags of the walk kernels.

    python tools/ab.py "name::ENV1=v,ENV2=v::-DFLAG=1 -DOTHER=2" ... [--wl=cit2,collab] [--steps=10] [--reps=2]
                       [--files=walk.hip,walk_pipe.hip] [--bench="--pairs 1024"]          (options take the --key=value form)

A variant with flags is compiled into /tmp and selected with SUBGACC_LIB (the shipped library is never touched); every
variant runs `bench.py --no-cpu-baseline --no-others` per workload and prints the stage times of the step."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "surel_plus_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off".split()


def build_variant(flags, files):
    tag = hashlib.sha1((flags + "|" + ",".join(files)).encode()).hexdigest()[:10]
    out = f"/tmp/libsubgacc_{tag}.so"
    if os.path.exists(out):
        return out
    objs = [os.path.join(CSRC, "build", f) for f in os.listdir(os.path.join(CSRC, "build"))
            if f.endswith(".o") and f[:-2] + ".hip" not in files]
    procs = []
    hooks = ["-include", os.path.join(ROOT, "tools", "dev_hooks.hpp")] if any(k in flags for k in ("SG_EXPERIMENT", "SJ_EXPERIMENT", "SG_STOP_AFTER")) else []
    for f in files:
        o = f"/tmp/{tag}_{f[:-4]}.o"
        procs.append((o, subprocess.Popen(["/opt/rocm/bin/hipcc"] + FLAGS + hooks + flags.split() + ["-c", os.path.join(CSRC, f), "-o", o])))
        objs.append(o)
    for o, p in procs:
        if p.wait() != 0:
            raise SystemExit(f"compile failed: {o}")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = dict(a[2:].split("=", 1) for a in sys.argv[1:] if a.startswith("--") and "=" in a)
    wls = opts.get("wl", "cit2,collab").split(",")
    steps, reps = opts.get("steps", "10"), int(opts.get("reps", "2"))
    files = opts.get("files", "walk.hip,walk_pipe.hip").split(",")
    extra = opts.get("bench", "").split()
    for v in args or ["base::::"]:
        name, env_s, flags = (v.split("::") + ["", ""])[:3]
        env = dict(os.environ)
        env.pop("SUBGACC_LIB", None)
        for kv in filter(None, env_s.split(",")):
            k, val = kv.split("=", 1)
            env[k] = val
        if flags.strip():
            env["SUBGACC_LIB"] = build_variant(flags.strip(), files)
        for w in wls:
            for _ in range(reps):
                p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", w, "--steps", steps, "--warmup", "2",
                                    "--no-cpu-baseline", "--no-others"] + extra, env=env, capture_output=True, text=True)
                try:
                    d = json.loads(p.stdout.strip().splitlines()[-1])       # the compact line: the two kernels' HIP-event times are in `roofline`
                    st = {"walk_sets": round(d["roofline"]["kernel_ms"], 4), "sjoin_fill": round(d["roofline"].get("join_kernel_ms") or 0.0, 4)}
                    print(f"[{name}] {w}: step {d['ms_per_step']:.4f} ms  {d['value'] / 1e6:.2f} M pairs/s  {st}", flush=True)
                except Exception:
                    print(f"[{name}] {w}: FAILED rc={p.returncode} {p.stderr[-400:]}", flush=True)


if __name__ == "__main__":
    main()
