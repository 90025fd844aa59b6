"""Developer tool (GPU box): soak test of the premise of tools/experiments/concurrent_runs.py (two sampling runs replaying at once).  One batch as a single run, then REPS times as two utterance
shards replayed side by side on the denoiser's two library handles; every shard result must equal the single run bit for bit.
  python tools/concurrency_soak.py <B> <steps> [R]      (REPS=<n> in the environment, default 50; R = product shape L=16)
Prints the repetitions in which an utterance differed and a summary line.  (History: with the layer-0 de-duplication experiment of
round 2, tools/experiments/l0_dedup/, this showed ~1 wrong utterance per 250 step pairs; without it none in 3 000+.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from convofusion_amd import scheduler  # noqa: E402
from convofusion_amd.distributed import shard_cfg_batch  # noqa: E402
from convofusion_amd.sampler import SamplingRun  # noqa: E402

B, N = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3 and sys.argv[3] == "R":
    bench.L, bench.S = 16, (24, 161, 24, 8, 1)
REPS = int(os.environ.get("REPS", "50"))
dev = torch.device("cuda", 0)
model = bench.make_model(dev)
L = bench.L
mems, masks = bench.make_inputs(B, dev, seed=1234)
sch = scheduler.DDPMScheduler(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                              variance_type="fixed_small", clip_sample=True)


def open_(a, b, side):
    m = [shard_cfg_batch(x, a, b, B) for x in mems]
    mk = {k: shard_cfg_batch(v, a, b, B) for k, v in masks.items()}
    return SamplingRun(model, sch, m, mk, b - a, L, 1000, guidance_scale=7.5, seed=0, first_utterance=a, side_engine=side)


with open_(0, B, False) as r:
    r.steps(N)
    full = r.read(close=True)
with open_(0, B, False) as r:
    r.steps(N)
    print("single run deterministic:", bool(torch.equal(full, r.read(close=True))))
h = (B + 1) // 2
bad_reps = 0
for rep in range(REPS):
    r0, r1 = open_(0, h, False), open_(h, B, True)
    for _ in range(N):
        r0.steps(1)
        r1.steps(1)
        if os.environ.get("SYNC_EACH"):     # experiment: no host run-ahead, the two graphs still overlap inside the step
            torch.cuda.synchronize()
    p0, p1 = r0.read(close=True), r1.read(close=True)
    bad0 = [int(i) for i in torch.nonzero((p0 - full[:h]).abs().amax(dim=(1, 2)) > 0).flatten()]
    bad1 = [int(i) for i in torch.nonzero((p1 - full[h:]).abs().amax(dim=(1, 2)) > 0).flatten()]
    if bad0 or bad1:
        bad_reps += 1
        cnt = [int(((p0[i] - full[i]).abs() > 0).sum()) for i in bad0] + [int(((p1[i] - full[h + i]).abs() > 0).sum()) for i in bad1]
        print("rep", rep, "utterances that differ: shard 0", bad0, " shard 1", bad1, " elements that differ per utterance (of", p0[0].numel(), "):", cnt)
print(f"concurrency soak: B={B} L={L} steps={N}: {bad_reps} of {REPS} repetitions differed ({REPS * N} step pairs)")

if os.environ.get("ORIGIN"):
    # where a difference starts: the single run's latents after every step, then concurrent shards compared step by step
    # (the per-step read synchronises, so the two graphs still start together inside every step)
    NS = int(os.environ["ORIGIN"])
    with open_(0, B, False) as r:
        traj = []
        for _ in range(NS):
            r.steps(1)
            traj.append(r.read())
    found = 0
    for rep in range(REPS):
        r0, r1 = open_(0, h, False), open_(h, B, True)
        for k in range(NS):
            r0.steps(1)
            r1.steps(1)
            p = torch.cat([r0.read(), r1.read()], 0)
            d = (p - traj[k]).abs()
            if float(d.max()) > 0:
                utt = [int(i) for i in torch.nonzero(d.amax(dim=(1, 2)) > 0).flatten()]
                u = utt[0]
                du = d[u]
                rows = [int(i) for i in torch.nonzero(du.amax(dim=1) > 0).flatten()]
                print(f"rep {rep} step {k}: utterances {utt}; utterance {u}: {int((du > 0).sum())} of {du.numel()} elements differ, max abs {float(du.max()):.3e}, "
                      f"median abs of differing {float(du[du > 0].median()):.3e}, latent rows that differ: {len(rows)} of {du.shape[0]} (first {rows[:8]})")
                found += 1
                break
        r0.close()
        r1.close()
    print(f"origin search: {found} of {REPS} repetitions diverged within {NS} steps")
