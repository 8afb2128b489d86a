"""Bench one configuration under a list of environment settings:
python tools/probe_env.py CONFIG FRAMES_IN_FLIGHT VAR=VALUE [VAR=VALUE ...] (each setting is one run)"""
import os, subprocess, sys, json
cfg, fif = sys.argv[1], sys.argv[2]
for setting in sys.argv[3:]:
    env = dict(os.environ)
    k, v = setting.split("=", 1)
    env[k] = v
    out = subprocess.run([sys.executable, "bench.py", "--config", cfg, "--frames-in-flight", fif, "--no-cpu"], env=env, capture_output=True, text=True, timeout=300)
    try:
        j = json.loads(out.stdout.strip().splitlines()[-1])
        print(setting, "fif", fif, "ms", j["ms_per_step"], flush=True)
    except Exception:
        print(setting, "failed", out.stderr[-400:], flush=True)
