"""EXPERIMENT (archived, not part of the product): one batch as TWO open sampling runs on one GPU.

Moved out of ``convofusion_amd.sampler`` in round 3: on this stack (ROCm 7.2, MI355X) two graphs replaying
concurrently return wrong latents for single utterances about once per 1 000 - 1 500 step pairs, so the
package must not offer the path.  ``tools/concurrency_soak.py`` and ``tools/shard2_experiment.py`` import it from here.

Utterance shards [0, B/2) and [B/2, B) run on the denoiser's two library handles, each with its own captured hipGraph, workspace
and stream, replayed side by side; utterances are independent and the Philox streams are keyed by global utterance id, so the
latents SHOULD be bit-identical to the single run's, and two half-size graphs fill each other's kernel tails
(71.6 -> 76.0 steps/s at B = 32, L = 196).
"""
import torch

from convofusion_amd.sampler import CFG_CHUNKS, SamplingRun


def _utterance_slice(t, a, b, total, chunks):
    """Utterances [a, b) of a chunk-major guidance batch [chunks * total, ...] (same rule as distributed.shard_cfg_batch)."""
    if t is None:
        return None
    v = t.reshape(chunks, total, *t.shape[1:])
    return v[:, a:b].reshape(chunks * (b - a), *t.shape[1:]).contiguous()


class ConcurrentRuns:
    """EXPERIMENTAL, not used by default.  One batch as TWO open sampling runs -- utterance shards [0, B/2) and [B/2, B) -- on the
    denoiser's two library handles, each with its own captured hipGraph, workspace and stream, replayed side by side.

    Utterances are independent and the Philox streams are keyed by global utterance id, so the latents should be bit-identical to
    the single run's, and two half-size graphs fill each other's kernel tails: 71.6 -> 76.0 steps/s at B = 32, L = 196.  BUT on this
    stack (ROCm 7.2, MI355X) two graphs replaying concurrently are not reliable: ``tools/concurrency_soak.py`` shows single utterances
    with wrong latents about once per 1 000 - 1 500 step pairs (fused and three-launch attention paths alike, the shard whose graph
    is launched first in a step more often; a device synchronisation after every step pair does not remove it; buffers of the two
    handles are disjoint, no kernel uses scratch memory).  The cause was not found in round 2, so nothing
    in the package selects this class; it is reachable only through ``tools/concurrency_soak.py`` and ``tools/shard2_experiment.py``
    (``sample(..., concurrent_shards=2)`` and ``bench.py --shards 2`` were removed in round 3).
    Same interface as ``SamplingRun`` for steps / read / close.
    """

    def __init__(self, denoiser, scheduler, encoder_hidden_states, cond_masks, B, L, num_inference_steps, guidance_scale=7.5,
                 guidance_chunks=CFG_CHUNKS, eta=0.0, init_latents=None, step_noise=None, seed=0, first_utterance=0, preseq=None,
                 dedup=True, skip_zero_weight_chunks=False):
        if B < 2:
            raise ValueError("two concurrent shards need at least two utterances")
        G = guidance_chunks
        if encoder_hidden_states[0].shape[0] != G * B:
            raise ValueError(f"conditioning batch is {encoder_hidden_states[0].shape[0]} rows, expected G*B = {G * B}")
        self.B, self.L = B, L
        self.runs = []
        cuts = (0, (B + 1) // 2, B)
        try:
            for k in range(2):
                a, b = cuts[k], cuts[k + 1]
                mems = [_utterance_slice(m, a, b, B, G) for m in encoder_hidden_states]
                masks = {n: _utterance_slice(v, a, b, B, G) for n, v in (cond_masks or {}).items()}
                self.runs.append(SamplingRun(
                    denoiser, scheduler, mems, masks, b - a, L, num_inference_steps, guidance_scale, G, eta,
                    None if init_latents is None else init_latents[a:b], None if step_noise is None else step_noise[:, a:b],
                    seed, first_utterance + a, None if preseq is None else preseq[a:b], dedup, skip_zero_weight_chunks,
                    side_engine=bool(k)))
        except Exception:
            self.close()
            raise
        self.N = self.runs[0].N
        self.open = True

    def steps(self, n):
        for _ in range(int(n)):      # one replay per shard and iteration: the two streams advance together
            for r in self.runs:
                r.steps(1)

    @property
    def position(self):
        return self.runs[0].position

    def read(self, close=False):
        out = torch.cat([r.read(close) for r in self.runs], dim=0)
        if close:
            self.open = False
        return out

    def close(self):
        for r in self.runs:
            r.close()
        self.open = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
