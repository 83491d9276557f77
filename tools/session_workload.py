"""One external-action form of one BASELINE config, windows of K steps -- the workload tools/gpu_profile_session.sh puts
under rocprofv3 (and a quick probe on its own).

    python tools/session_workload.py --config 2 --form steps|session|lockstep|step_launches [--K 20] [--windows 40]

config 2 = ta01 x 4096 random, 3 = ta41 x 16384 SPT, 4 = synthetic 50x20 x 8192 random, 0 = ta01 x 65536 random (headline).
Forms: steps = jss_steps (K steps per launch); session = step session, K steps posted per wait; lockstep = step session,
one fused post + wait launch per step; step_launches = K x jss_step launches (the form the others replace).  Actions are a
recorded behaviour trajectory, resident in HBM.  Prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--form", default="steps")
    ap.add_argument("--K", type=int, default=20)
    ap.add_argument("--windows", type=int, default=40)
    ap.add_argument("--slots", type=int, default=0)
    args = ap.parse_args()
    import torch
    from jssenv_amd import BatchedJssEnv, builtin_instance
    from jssenv_amd.instances import synthetic_packed
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    src, batch, policy, J, M = {0: ("ta01", 65536, "random", 15, 15), 2: ("ta01", 4096, "random", 15, 15),
                                3: ("ta41", 16384, "SPT", 30, 20), 4: (None, 8192, "random", 50, 20)}[args.config]
    inst = synthetic_packed(batch, 50, 20) if src is None else builtin_instance(src)
    env = BatchedJssEnv(inst, batch=batch, device=dev, seed=0)
    env.reset()
    ids = torch.arange(batch, device=dev) % 16                 # spread the episode phases like bench.py does
    skip = torch.full((batch,), -1, dtype=torch.int32, device=dev)
    for r in range(15):
        for _ in range(16):
            env.step(torch.where(ids > r, env.policy(policy), skip))
    env.rollout(policy, n_iter=64)
    K, W = args.K, args.windows
    acts = env.trajectory(policy, steps=(W + 1) * K, record=("action",))["action"]
    counts = (acts >= 0).view(W + 1, K * batch).sum(1).cpu().tolist()
    snap = env._arena.clone(), env.solution.clone()             # (taken after the recording: the replay below starts where it ended
    cur = torch.cuda.current_stream(dev)                         #  -- irrelevant for timing, the actions stay legal-or-flagged either way)
    rows = []

    def run(issue):
        for w in range(W + 1):
            cur.synchronize()
            t0 = time.perf_counter()
            issue(w)
            cur.synchronize()
            rows.append(counts[w] / (time.perf_counter() - t0))

    # replay from the state the trajectory started in
    env2 = BatchedJssEnv(inst, batch=batch, device=dev, seed=0)
    env2.reset()
    for r in range(15):
        for _ in range(16):
            env2.step(torch.where(ids > r, env2.policy(policy), skip))
    env2.rollout(policy, n_iter=64)
    del env, snap
    env = env2
    extra = {}
    if args.form == "steps":                                     # every step's outputs recorded, like the session writes them
        rec = ("real_obs", "action_mask", "reward", "done")
        Jm = env.jmax
        bufs = {"real_obs": torch.zeros((K, batch, Jm, 7), dtype=torch.float32, device=dev),
                "action_mask": torch.zeros((K, batch, Jm + 1), dtype=torch.uint8, device=dev),
                "reward": torch.zeros((K, batch), dtype=torch.float32, device=dev),
                "done": torch.zeros((K, batch), dtype=torch.uint8, device=dev)}
        run(lambda w: env.steps(acts[w * K:(w + 1) * K], record=rec, buffers=bufs))
    elif args.form == "steps_bare":                              # reward / done per step, observation of the last step only
        run(lambda w: env.steps(acts[w * K:(w + 1) * K]))
    elif args.form == "step_launches":
        def launches(w):
            for k in range(K):
                env.step(acts[w * K + k])
        run(launches)
    else:
        depth = K if args.form == "session" else 1
        with env.session(depth=depth, slots=args.slots) as s:
            if args.form == "session":
                run(lambda w: (s.post(acts[w * K:(w + 1) * K]), s.wait()))
            else:
                def lock(w):
                    for k in range(K):
                        s.step(acts[w * K + k])
                run(lock)
        extra = s.host_status()
    torch.cuda.synchronize()
    err = int(env.err.max().item())
    r = sorted(rows[1:])
    med = r[len(r) // 2]
    alg = 89 * J + 10 * M + 40
    print(json.dumps({"config": args.config, "form": args.form, "K": K, "windows": len(r), "env_steps_per_s": med, "min": r[0], "max": r[-1],
                      "us_per_step": 1e6 * sum(counts[1:]) / len(r) / med / K, "roofline_frac": med * alg / 8e12,
                      "err_flags": err, **extra}))


if __name__ == "__main__":
    main()
