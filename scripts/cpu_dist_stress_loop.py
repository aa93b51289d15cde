"""Repeat the two-rank CPU twin of bench.py's N > 1 loop (tests/mockhip/dist_pipeline_stress.py: two processes against the stand-in
HIP runtime and the stand-in collective) until it fails or REPS repetitions went through.  Every repetition draws its own exchange
form, stream-synchronisation delay, planting period and step count, so that the two ranks' threads interleave differently each
time.  VERDICT round 5, item 2: "500 clean repetitions of the CPU twin".
    python scripts/cpu_dist_stress_loop.py [REPS] [LOG]
A failing repetition's output (both ranks) is appended to LOG and the loop goes on; the summary line counts them."""
import os
import random
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 500
LOG = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "cpu_dist_stress_loop.log")


def build(d):
    mockhip = os.path.join(d, "libmockhip.so")
    subprocess.run(["gcc", "-O1", "-w", "-fPIC", "-shared", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "mockhip", "mockhip.c"), "-o", mockhip], check=True)
    rccl_dir = os.path.join(d, "rccl")
    os.makedirs(rccl_dir)
    subprocess.run(["g++", "-O1", "-w", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-Wl,-soname,librccl.so.1",
                    os.path.join(ROOT, "tests", "mockrccl", "mockrccl.cpp"), "-o", os.path.join(rccl_dir, "librccl.so.1")], check=True)
    return mockhip, rccl_dir


def main():
    os.makedirs(os.path.dirname(LOG), exist_ok=True)
    d = tempfile.mkdtemp(prefix="nrtgpu_stress_")
    mockhip, rccl_dir = build(d)
    rng = random.Random(int(os.environ.get("SEED", "6")))
    bad = 0
    t0 = time.time()
    with open(LOG, "a") as log:
        for rep in range(REPS):
            mode = rng.choice(["alltoall", "allgather"])
            sync_us = rng.choice([0, 0, 20, 50, 200, 1000])
            plant = rng.choice([0, 2, 3, 5, 7])
            steps = rng.choice([40, 80, 120])
            sync_dir = tempfile.mkdtemp(prefix="nrtgpu_dist2s_")
            env = dict(os.environ, LD_PRELOAD=mockhip, LD_LIBRARY_PATH=rccl_dir + ":" + os.environ.get("LD_LIBRARY_PATH", ""), MOCKHIP_SYNC_US=str(sync_us),
                       PLANT=str(plant), WATCHDOG="120")
            env.pop("NRTGPU_LIB_PATH", None)
            procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mockhip", "dist_pipeline_stress.py"), str(r), "2", sync_dir, mode, str(steps)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
            outs = []
            for p in procs:
                try:
                    o, _ = p.communicate(timeout=200)
                except subprocess.TimeoutExpired:
                    p.kill()
                    o, _ = p.communicate()
                    o += "\n[killed: timeout]"
                outs.append(o)
            ok = all(p.returncode == 0 for p in procs) and all("done" in o for o in outs)
            shutil.rmtree(sync_dir, ignore_errors=True)
            if not ok:
                bad += 1
                log.write(f"=== repetition {rep}: mode {mode} sync_us {sync_us} plant {plant} steps {steps}: FAILED\n")
                for r, o in enumerate(outs):
                    log.write(f"--- rank {r} (rc {procs[r].returncode})\n{o[-6000:]}\n")
                log.flush()
            if rep % 25 == 24:
                print(f"{rep + 1} repetitions, {bad} failed, {time.time() - t0:.0f} s", flush=True)
        log.write(f"=== {REPS} repetitions, {bad} failed, {time.time() - t0:.0f} s (seed {os.environ.get('SEED', '6')})\n")
    shutil.rmtree(d, ignore_errors=True)
    print(f"{REPS} repetitions, {bad} failed")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
