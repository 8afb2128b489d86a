"""A/B of alternative builds of the library: python tools/ab_libs.py CONFIG LIB [LIB...]
("default" = the in-tree build; extra bench arguments in AB_ARGS).  Runs pytest-free: bench only."""
import os, subprocess, sys, json
cfg = sys.argv[1]
for lib in sys.argv[2:]:
    env = dict(os.environ)
    if lib != "default":
        env["RENDERER_AMD_LIB"] = os.path.abspath(lib)
    for rep in range(2):
        out = subprocess.run([sys.executable, "bench.py", "--config", cfg, "--steps", "200", "--warmup", "20", "--frames-per-step", "1", "--no-cpu"] + os.environ.get("AB_ARGS", "").split(),
                             env=env, capture_output=True, text=True, timeout=300)
        try:
            j = json.loads(out.stdout.strip().splitlines()[-1])
            print(lib, "cfg", cfg, "ms", j["ms_per_step"], j["stage_ms"], flush=True)
        except Exception:
            print(lib, "failed", out.stderr[-800:], flush=True)
            sys.exit(1)
