"""Data-side formats either side of the hot path — host-side Python, no GPU work:

* the SpeechBrain CSV manifest the reference's data preparation writes (``ID,duration,wav,spk_id,wrd``,
  librispeech_prepare.py:136, 646-691);
* duration-bucketed dynamic batching (the role of speechbrain's DynamicBatchSampler configured at
  hparams/CTC/conmamba_large.yaml:108-117, 254-270 and instantiated at train_CTC.py:991-1010);
* CTC greedy decoding (train_CTC.py:305-310 -> speechbrain.decoders.ctc_greedy_decode);
* checkpoint averaging (train_CTC.py:479-488 -> speechbrain average_checkpoints).

speechbrain is not in the reference tree nor installed: these restate its documented behaviour (parity unpinned) and
are pinned by the properties tests/test_dataio.py states."""
from __future__ import annotations

import csv
import math
import random
from typing import Dict, Iterable, Iterator, List, Optional, Sequence

import torch

CSV_HEADERS = ["ID", "duration", "wav", "spk_id", "wrd"]


def write_manifest(rows: Iterable[Dict], path: str) -> None:
    """rows: dicts with the CSV_HEADERS keys -> the CSV the reference's create_csv writes (QUOTE_MINIMAL, header row)."""
    with open(path, "w", newline="") as f:
        w = csv.writer(f, delimiter=",", quotechar='"', quoting=csv.QUOTE_MINIMAL)
        w.writerow(CSV_HEADERS)
        for r in rows:
            w.writerow([r["ID"], str(r["duration"]), r["wav"], r["spk_id"], r["wrd"]])


def read_manifest(path: str) -> List[Dict]:
    """-> list of {"ID", "duration" (float, seconds), "wav", "spk_id", "wrd"} in file order."""
    with open(path, newline="") as f:
        rd = csv.DictReader(f)
        missing = [h for h in CSV_HEADERS if h not in (rd.fieldnames or [])]
        if missing:
            raise ValueError(f"{path}: manifest lacks column(s) {missing}")
        return [dict(r, duration=float(r["duration"])) for r in rd]


class DurationBucketBatchSampler:
    """Batches of utterance indices whose padded length (longest member x members) stays under ``max_batch_length``
    seconds and ``max_batch_ex`` members; utterances are grouped into ``num_buckets`` duration buckets first so that a
    batch pads little.  ``batch_ordering``: 'random' (reshuffled per epoch, seeded), 'ascending' or 'descending' by
    the batch's longest member.  ``shuffle`` re-draws the membership inside each bucket every epoch."""

    def __init__(self, durations: Sequence[float], max_batch_length: float, num_buckets: int = 200, shuffle: bool = False,
                 max_batch_ex: Optional[int] = None, batch_ordering: str = "random", seed: int = 0, rank: int = 0, world: int = 1):
        if batch_ordering not in ("random", "ascending", "descending"):
            raise ValueError(f"batch_ordering {batch_ordering!r}")
        if any(d > max_batch_length for d in durations):
            raise ValueError("an utterance is longer than max_batch_length")
        self.durations, self.max_len, self.max_ex = list(durations), float(max_batch_length), max_batch_ex
        self.num_buckets, self.shuffle, self.ordering, self.seed = max(1, num_buckets), shuffle, batch_ordering, seed
        self.rank, self.world, self.epoch = rank, world, 0
        self._batches = self._make()

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch
        if self.shuffle or self.ordering == "random":
            self._batches = self._make()

    def _make(self) -> List[List[int]]:
        rng = random.Random(self.seed + self.epoch)
        order = sorted(range(len(self.durations)), key=lambda i: self.durations[i])
        per = math.ceil(len(order) / self.num_buckets) if order else 0
        batches: List[List[int]] = []
        for b0 in range(0, len(order), max(per, 1)):
            bucket = order[b0:b0 + per]
            if self.shuffle:
                rng.shuffle(bucket)
            cur: List[int] = []
            longest = 0.0
            for i in bucket:
                new_longest = max(longest, self.durations[i])
                if cur and (new_longest * (len(cur) + 1) > self.max_len or (self.max_ex and len(cur) >= self.max_ex)):
                    batches.append(cur)
                    cur, new_longest = [], self.durations[i]
                cur.append(i)
                longest = new_longest
            if cur:
                batches.append(cur)
        if self.ordering == "random":
            rng.shuffle(batches)
        else:
            batches.sort(key=lambda bt: max(self.durations[i] for i in bt), reverse=self.ordering == "descending")
        return batches

    def _shard(self) -> List[List[int]]:
        # data parallel: rank r takes batches r, r + world, ... of the common list (every rank builds the same list).
        # Every rank must see the SAME number of batches -- each stepping micro-batch ends in a collective, and a rank
        # with one batch more would wait for peers that never arrive -- so the list is padded to a multiple of the world
        # size by repeating its leading batches (what torch's DistributedSampler does with drop_last=False).
        batches = self._batches
        if self.world > 1 and batches and len(batches) % self.world:
            pad = self.world - len(batches) % self.world
            batches = batches + [batches[i % len(batches)] for i in range(pad)]
        return batches[self.rank::self.world]

    def __iter__(self) -> Iterator[List[int]]:
        return iter(self._shard())

    def __len__(self) -> int:
        return len(self._shard())


def ctc_greedy_decode(log_probs: torch.Tensor, wav_lens: torch.Tensor, blank_id: int = 0) -> List[List[int]]:
    """(batch, steps, vocab) log-probabilities + relative lengths -> per utterance the arg-max path with repeats collapsed
    and blanks removed (train_CTC.py:305-310)."""
    steps = log_probs.shape[1]
    best = log_probs.argmax(dim=-1).cpu()
    out = []
    for row, rel in zip(best, wav_lens.cpu().tolist()):
        n = int(round(rel * steps))
        seq, prev = [], None
        for tok in row[:n].tolist():
            if tok != prev and tok != blank_id:
                seq.append(tok)
            prev = tok
        out.append(seq)
    return out


def average_checkpoints(state_dicts: Sequence[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    """Element-wise mean of the floating-point entries of several state_dicts with identical keys (integer buffers are
    taken from the first), as the reference averages its best checkpoints before evaluation (train_CTC.py:479-488)."""
    if not state_dicts:
        raise ValueError("no checkpoints to average")
    keys = list(state_dicts[0].keys())
    for sd in state_dicts[1:]:
        if list(sd.keys()) != keys:
            raise ValueError("checkpoints have different keys")
    avg = {}
    for k in keys:
        first = state_dicts[0][k]
        if torch.is_floating_point(first):
            acc = first.detach().to(torch.float64).clone()
            for sd in state_dicts[1:]:
                acc += sd[k].detach().to(torch.float64)
            avg[k] = (acc / len(state_dicts)).to(first.dtype)
        else:
            avg[k] = first.clone()
    return avg
