"""HBM bytes per env-step of arl_env_step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/env_step_pmc.sh).
FETCH_SIZE is reported raw and doubled (MI355X_MICROARCH.md: gfx950 tallies the 128-byte requests of wide coalesced reads
at 64 B); which one applies to this kernel's loads is calibrated on its known reads: the older three frames of the stack
(24 960 B per env-step from HBM -- the 5 GB rollout buffer is no cache's tenant) plus whatever part of the two raw
frames (67 200 B per env-step out of a 2 MB frame bank that lives in L2 / the Infinity Cache) reaches the fabric counters.
usage: env_step_pmc.py fetch.csv write.csv n_envs"""
import csv, hashlib, json, os, sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "accel_rl_amd", "csrc")


def env_sources_sha1():
    """sha1 over the env step kernel's sources (env.hip + the device pieces it shares with serve_step.hip)"""
    h = hashlib.sha1()
    for name in ("env.hip", "env_dev.h"):
        h.update(open(os.path.join(CSRC, name), "rb").read())
    return h.hexdigest()


def med(path, counter):
    v = sorted(float(r["Counter_Value"]) for r in csv.DictReader(open(path))
               if "env_step_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter)
    return v[len(v) // 2], len(v)


n = int(sys.argv[3])
f_kb, k = med(sys.argv[1], "FETCH_SIZE")
w_kb, _ = med(sys.argv[2], "WRITE_SIZE")
algo = 75520 + 33295 + 4 * 4
fetch_raw, write = f_kb * 1024 / n, w_kb * 1024 / n
print(json.dumps(dict(
    n_envs=n, kernel="env_step_kernel (single_write)", launches=k, FETCH_SIZE_KB=f_kb, WRITE_SIZE_KB=w_kb,
    fetch_bytes_per_env_step_raw=round(fetch_raw, 1), fetch_bytes_per_env_step_doubled=round(2 * fetch_raw, 1),
    write_bytes_per_env_step=round(write, 1),
    algorithmic_bytes_per_env_step=algo,
    kernel_reads_per_env_step=dict(raw_frames_from_the_2MB_bank=67200, older_three_frames_of_the_stack=24960),
    kernel_writes_per_env_step=dict(stacked_row_incl_the_new_frame=33280, scalars_and_prob=15 + 4 * 4),
    hbm_bytes_per_env_step=round(2 * fetch_raw + write, 1),
    traffic_over_algorithmic=round((2 * fetch_raw + write) / algo, 3),
    env_hip_sha1=env_sources_sha1(),
    note="hbm_bytes_per_env_step = 2 x FETCH_SIZE + WRITE_SIZE (the guide's gfx950 correction for 16-byte-per-lane "
         "coalesced reads, which is what the kernel's frame and stack loads are); algorithmic bytes = preprocess + "
         "stack 75 520 + rollout store 33 295 + 4 A (SURVEY 8d).  Traffic BELOW the algorithmic figure: the raw frames come "
         "out of a 2 MB bank that stays in L2 / the Infinity Cache, and the new frame travels inside the stacked row"), indent=1))
