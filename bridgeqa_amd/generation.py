"""Beam-search decoding for the answer decoder -- what `self.text_decoder.generate(...)` does in the reference's
open-ended inference (models/blip_vqa_3d.py:394-417: `num_beams = 10`, `max_length = 20`, `min_length = 1`, eos = [SEP]).

PARITY UNPINNED.  The reference inherits `generate` from HuggingFace `transformers` (GenerationMixin.generate ->
beam_search with a BeamSearchScorer), a third-party dependency that is NOT in the reference tree and NOT pinned by its
requirements.txt; models/med.py's header names v4.15.0, which is the version whose PUBLISHED algorithm is restated here
(generation_utils.py `beam_search`, generation_beam_search.py `BeamSearchScorer` / `BeamHypotheses`, with the defaults
`generate` passes: length_penalty 1.0, early_stopping False, one returned sequence, no n-gram / repetition processors;
MinLengthLogitsProcessor(min_length, eos)).  The image's transformers (5.x) no longer gives the reference's decoder a
`generate` at all, so no vector of the reference's own exists to pin this to; the tests check it against an exhaustive
search and against greedy decoding instead (tests/test_generate_gpu.py).  What IS taken from the reference tree:
`prepare_inputs_for_generation` (med.py:1447-1464: all-ones attention mask, only the last token once a cache exists,
`is_decoder=True`) and `_reorder_cache` (:1466-1470: index_select of every cached self-attention K/V by the beam index).

Semantics kept on purpose: the encoder states are given per BEAM SLOT (the caller repeats them) and are NOT reordered
with the beams -- the reference puts the 2D question states into slots 0-4 and the 3D ones into slots 5-9 of every
sample (blip_vqa_3d.py:396-401), so a hypothesis can move between the two contexts while its cached self-attention keys
stay what they were.

Device-side: one decoder call per step for all B x beams slots; the top-2k selection, the beam bookkeeping of OPEN beams
(which candidate becomes which next beam) and the cache gather are tensor ops; only the finished-hypothesis heaps -- a
few Python floats and token lists per sample -- live on the host, fed by one small copy per step (the reference's scorer
does the same walk with a `.item()` per candidate).
"""
import torch


class BeamHypotheses(object):
    """generation_beam_search.py BeamHypotheses: the num_beams best finished hypotheses of one sample, scored by
    sum_logprobs / len ** length_penalty"""

    def __init__(self, num_beams, length_penalty, early_stopping):
        self.num_beams, self.length_penalty, self.early_stopping = num_beams, length_penalty, early_stopping
        self.beams = []
        self.worst_score = 1e9

    def __len__(self):
        return len(self.beams)

    def add(self, hyp, sum_logprobs):
        score = sum_logprobs / (len(hyp) ** self.length_penalty)
        if len(self) < self.num_beams or score > self.worst_score:
            self.beams.append((score, hyp))
            if len(self) > self.num_beams:
                ranked = sorted([(s, idx) for idx, (s, _) in enumerate(self.beams)])
                del self.beams[ranked[0][1]]
                self.worst_score = ranked[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs, cur_len):
        if len(self) < self.num_beams:
            return False
        if self.early_stopping:
            return True
        return self.worst_score >= best_sum_logprobs / cur_len ** self.length_penalty


@torch.no_grad()
def beam_search(step_fn, reorder_fn, input_ids, num_beams, max_length, eos_token_id, pad_token_id, min_length=0,
                length_penalty=1.0, early_stopping=False):
    """input_ids (B * num_beams, L0) (every sample's prompt repeated num_beams times, as generate's
    _expand_inputs_for_generation leaves it); step_fn(input_ids, past) -> (logits of the last position (B * num_beams, V),
    past); reorder_fn(past, beam_idx) -> past.  Returns (sequences (B, <= max_length) padded with pad_token_id, scores (B,))."""
    dev = input_ids.device
    BB, cur_len = input_ids.shape
    B = BB // num_beams
    hyps = [BeamHypotheses(num_beams, length_penalty, early_stopping) for _ in range(B)]
    done = [False] * B
    beam_scores = torch.zeros(B, num_beams, dtype=torch.float32, device=dev)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(-1)
    base = (torch.arange(B, device=dev) * num_beams).unsqueeze(1)
    past = None
    while True:
        logits, past = step_fn(input_ids, past)
        scores = torch.log_softmax(logits.float(), dim=-1)
        if cur_len < min_length:                                   # MinLengthLogitsProcessor
            scores[:, eos_token_id] = -float("inf")
        V = scores.shape[-1]
        scores = (scores + beam_scores[:, None]).view(B, num_beams * V)
        top_scores, top = torch.topk(scores, 2 * num_beams, dim=1, largest=True, sorted=True)
        top_beam, top_tok = torch.div(top, V, rounding_mode="floor"), top % V
        # ---- BeamSearchScorer.process, open beams on the device: the first num_beams non-eos candidates in rank order
        is_eos = top_tok == eos_token_id
        order = torch.cumsum((~is_eos).long(), 1)
        take = (~is_eos) & (order <= num_beams)
        sel = torch.nonzero(take)[:, 1].view(B, num_beams)         # 2 * num_beams candidates always hold num_beams non-eos
        next_scores = torch.gather(top_scores, 1, sel)
        next_tok = torch.gather(top_tok, 1, sel)
        next_idx = torch.gather(top_beam, 1, sel) + base
        # ---- finished hypotheses: eos candidates ranked inside the first num_beams (host heaps; one small copy per step)
        eos_rank_ok = is_eos & (torch.arange(2 * num_beams, device=dev)[None, :] < num_beams)
        h_eos, h_scores, h_beam = eos_rank_ok.cpu(), top_scores.cpu(), (top_beam + base).cpu()
        ids_host = input_ids.cpu() if bool(h_eos.any()) else None
        for b in range(B):
            if done[b]:
                continue
            for j in torch.nonzero(h_eos[b])[:, 0].tolist():
                hyps[b].add(ids_host[int(h_beam[b, j])].tolist(), float(h_scores[b, j]))
            done[b] = hyps[b].is_done(float(h_scores[b].max()), cur_len)
        if any(done):                                              # a finished sample keeps decoding padding (as HF does)
            dmask = torch.tensor(done, device=dev)
            next_scores = torch.where(dmask[:, None], torch.zeros_like(next_scores), next_scores)
            next_tok = torch.where(dmask[:, None], torch.full_like(next_tok, pad_token_id), next_tok)
            next_idx = torch.where(dmask[:, None], base.expand_as(next_idx), next_idx)
        beam_scores = next_scores.reshape(-1)
        beam_idx = next_idx.reshape(-1)
        input_ids = torch.cat([input_ids[beam_idx], next_tok.reshape(-1, 1)], dim=-1)
        past = reorder_fn(past, beam_idx)
        cur_len += 1
        if all(done) or cur_len >= max_length:
            break
    # ---- BeamSearchScorer.finalize: open beams of unfinished samples join the heaps; best hypothesis per sample
    ids_host, sc_host = input_ids.cpu(), beam_scores.cpu()
    best, best_scores = [], []
    for b in range(B):
        if not done[b]:
            for k in range(num_beams):
                hyps[b].add(ids_host[b * num_beams + k].tolist(), float(sc_host[b * num_beams + k]))
        score, hyp = sorted(hyps[b].beams, key=lambda x: x[0])[-1]
        best.append(hyp)
        best_scores.append(score)
    sent_max = min(max(len(h) for h in best) + 1, max_length)
    out = torch.full((B, sent_max), pad_token_id, dtype=torch.long)
    for b, h in enumerate(best):
        out[b, :len(h)] = torch.tensor(h, dtype=torch.long)
        if len(h) < max_length:
            out[b, len(h)] = eos_token_id
    return out.to(dev), torch.tensor(best_scores, dtype=torch.float32, device=dev)
