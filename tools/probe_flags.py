"""Times the stages of one configuration under each RND_DEBUG probe mask (results are wrong
under a probe; only durations mean anything).  Usage: python tools/probe_flags.py CONFIG MASK [MASK...]

The probes exist only in a build with -DRND_PROBES:
    EXTRA=-DRND_PROBES tools/build_variant.sh probes && RENDERER_AMD_LIB=build/lib_probes.so python tools/probe_flags.py 1 0 1 2"""
import os, subprocess, sys, json
cfg = sys.argv[1]
for m in sys.argv[2:]:
    env = dict(os.environ, RND_DEBUG=m)
    out = subprocess.run([sys.executable, "bench.py", "--config", cfg, "--steps", "200", "--warmup", "20", "--frames-per-step", "1", "--no-cpu"],
                         env=env, capture_output=True, text=True, timeout=300)
    try:
        j = json.loads(out.stdout.strip().splitlines()[-1])
        print("mask", m, "ms", j["ms_per_step"], j["stage_ms"], flush=True)
    except Exception as e:
        print("mask", m, "failed", out.stderr[-500:], flush=True)
