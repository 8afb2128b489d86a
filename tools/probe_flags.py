"""Times the stages of one configuration under each RND_DEBUG probe mask (results are wrong
under a probe; only durations mean anything).  Usage: python tools/probe_flags.py CONFIG MASK [MASK...]"""
import os, subprocess, sys, json
cfg = sys.argv[1]
for m in sys.argv[2:]:
    env = dict(os.environ, RND_DEBUG=m)
    out = subprocess.run([sys.executable, "bench.py", "--config", cfg, "--steps", "200", "--warmup", "20", "--no-cpu"],
                         env=env, capture_output=True, text=True, timeout=300)
    try:
        j = json.loads(out.stdout.strip().splitlines()[-1])
        print("mask", m, "ms", j["ms_per_step"], j["stage_ms"], flush=True)
    except Exception as e:
        print("mask", m, "failed", out.stderr[-500:], flush=True)
