"""Data-side formats: CSV manifest round trip, duration-bucketed batching invariants, CTC greedy
decoding, checkpoint averaging.  CPU only."""
import torch

from mamba_asr_amd import dataio


def test_manifest_round_trip(tmp_path):
    rows = [{"ID": "1089-134686-0000", "duration": 5.535, "wav": "/data/a,b.flac", "spk_id": "1089-134686",
             "wrd": 'HE SAID "NO", TWICE'},
            {"ID": "x-1", "duration": 12.0, "wav": "/data/x.flac", "spk_id": "x", "wrd": "PLAIN WORDS"}]
    p = tmp_path / "train.csv"
    dataio.write_manifest(rows, str(p))
    assert p.read_text().splitlines()[0] == "ID,duration,wav,spk_id,wrd"
    back = dataio.read_manifest(str(p))
    assert back == rows and isinstance(back[0]["duration"], float)


def test_bucket_sampler_invariants_and_sharding():
    g = torch.Generator().manual_seed(0)
    dur = torch.exp(torch.randn(2000, generator=g) * 0.6 + 2.5).clamp(1.3, 35.0).tolist()       # LibriSpeech-like seconds
    s = dataio.DurationBucketBatchSampler(dur, max_batch_length=850, num_buckets=200, max_batch_ex=128, seed=3)
    batches = list(s)
    seen = sorted(i for b in batches for i in b)
    assert seen == list(range(2000))                                   # every utterance exactly once
    for b in batches:
        assert len(b) <= 128 and max(dur[i] for i in b) * len(b) <= 850 + 1e-9
    pad = sum(max(dur[i] for i in b) * len(b) for b in batches) / sum(dur)
    assert pad < 1.1                                                    # bucketing keeps padding under 10 %
    s.set_epoch(1)
    assert list(s) != batches and sorted(i for b in s for i in b) == list(range(2000))    # reshuffled, still a partition
    asc = dataio.DurationBucketBatchSampler(dur, 850, num_buckets=50, batch_ordering="ascending")
    longest = [max(dur[i] for i in b) for b in asc]
    assert longest == sorted(longest)
    shards = [dataio.DurationBucketBatchSampler(dur, 850, seed=3, rank=r, world=4) for r in range(4)]
    union = sorted(set(i for sh in shards for b in sh for i in b))
    assert union == list(range(2000))
    # every rank yields the SAME number of batches (each stepping micro-batch ends in a collective): the common list is
    # padded with its leading batches when it does not divide by the world size
    assert len({len(sh) for sh in shards}) == 1 and len({len(list(sh)) for sh in shards}) == 1
    for world in (2, 3, 4, 7):
        sh = [dataio.DurationBucketBatchSampler(dur[:331], 850, seed=1, rank=r, world=world) for r in range(world)]
        assert len({len(list(x)) for x in sh}) == 1
        assert sorted(set(i for x in sh for b in x for i in b)) == list(range(331))


def test_ctc_greedy_decode_collapses_and_drops_blanks():
    path = [[0, 3, 3, 0, 3, 4, 4, 0, 0, 5], [7, 7, 0, 7, 0, 0, 2, 2, 2, 9]]
    logp = torch.full((2, 10, 12), -10.0)
    for b, row in enumerate(path):
        for t, k in enumerate(row):
            logp[b, t, k] = 0.0
    got = dataio.ctc_greedy_decode(logp, torch.tensor([1.0, 0.6]), blank_id=0)
    assert got == [[3, 3, 4, 5], [7, 7]]                                # second utterance: only its first 6 steps count


def test_average_checkpoints():
    a = {"w": torch.tensor([1.0, 3.0]), "steps": torch.tensor(5)}
    b = {"w": torch.tensor([3.0, 5.0]), "steps": torch.tensor(9)}
    avg = dataio.average_checkpoints([a, b])
    torch.testing.assert_close(avg["w"], torch.tensor([2.0, 4.0]))
    assert int(avg["steps"]) == 5 and avg["w"].dtype == torch.float32
