"""Shared test helpers: rebuild the synthetic weights / inputs the golden fixtures were made from."""
import json
import os

import numpy as np
import torch

from candidate_reranking_cir_amd import config as cfgmod, synthetic, weights

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FULL_BERT = dict(hidden_size=768, num_attention_heads=12, num_hidden_layers=12, intermediate_size=3072,
                 layer_norm_eps=1e-12, vocab_size=30524, max_position_embeddings=512, encoder_width=768)


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def geometry(bert_cfg: dict, vit_cfg: dict):
    return cfgmod.BertGeometry.from_dict(bert_cfg), cfgmod.VitGeometry(**vit_cfg)


def fixture_images(z, ids, size):
    """The images a fixture was generated from: structured `scene_images` when the fixture says so, i.i.d. noise otherwise."""
    if "image_kind" in z.files and str(z["image_kind"]) == "scene":
        return synthetic.scene_images(ids, size)
    return synthetic.images(ids, size)


_SD_CACHE = {}


def state_dicts(g, v, seed, profile):
    """Synthesised (stage-II, stage-I) state dicts; the last few (geometry, seed, profile) combinations stay cached - a full-size
    pair is 0.5 G parameters of CPU random numbers (~25 s), and a dozen tests build the same models.  Callers only read them
    (`load_state_dict` copies)."""
    key = (repr(g), repr(v), int(seed), str(profile))
    if key not in _SD_CACHE:
        if len(_SD_CACHE) >= 3:
            _SD_CACHE.pop(next(iter(_SD_CACHE)))
        _SD_CACHE[key] = (weights.synth_state_dict(weights.nlvr_param_spec(g, v), seed, profile),
                          weights.synth_state_dict(weights.retrieval_param_spec(g, v), seed + 1, profile))
    return _SD_CACHE[key]


def tiny_setup():
    z = load("tiny_loop.npz")
    g, v = geometry(json.loads(str(z["bert_cfg"])), json.loads(str(z["vit_cfg"])))
    sd2, sd1 = state_dicts(g, v, int(z["seed"]), str(z["profile"]))
    return z, g, v, sd2, sd1


def fiq_caption(pair):
    """validate_stage2.py:97-100: 'Cap1 and cap2' with '.?, ' stripped and the first capitalised."""
    a, b = str(pair[0]), str(pair[1])
    return f"{a.strip('.?, ').capitalize()} and {b.strip('.?, ')}"


def tokenize(texts):
    enc = synthetic.HashTokenizer()(list(texts))
    return enc.input_ids, enc.attention_mask


def grad_sample_index(numel: int, n: int = 64) -> np.ndarray:
    """The positions of a flattened gradient kept by tests/golden/train768.npz (oracle/make_golden.py: grad_sample_index)."""
    return (np.arange(n, dtype=np.int64) * 2654435761 + 12345) % numel


class DuckFIQ:
    """A FashionIQ 'relative' val dataset with a top-K file, as the reference's callers see it (data_utils.py:166-179: `K`,
    `K_labels`, `K_sorted_index_names`, `dress_types`; items as data_utils.py:204-208) - built from fixture arrays."""

    def __init__(self, names, refs, targets, captions, cand_idx, labels):
        self.names, self.refs, self.targets, self.captions = list(names), refs, targets, captions
        self.K_sorted_index_names = np.array(self.names)[cand_idx]
        self.K_labels, self.K = labels, cand_idx.shape[1]
        self.dress_types, self.split = ["dress"], "val"

    def __len__(self):
        return len(self.refs)

    def __getitem__(self, i):
        return (self.names[self.refs[i]], self.names[self.targets[i]], [str(c) for c in self.captions[i]],
                self.K_sorted_index_names[i].tolist(), self.K_labels[i])


class DuckCIRR(DuckFIQ):
    """CIRR 'relative' val items (data_utils.py:332-336): the 6 subset members INCLUDE the reference image."""

    def __init__(self, names, refs, targets, captions, cand_idx, labels, groups, ref_slot=0):
        super().__init__(names, refs, targets, captions, cand_idx, labels)
        self.groups, self.ref_slot = groups, ref_slot
        self.K_group_labels = np.zeros((len(refs), 5), dtype=bool)

    def __getitem__(self, i):
        members = [self.names[j] for j in self.groups[i]]
        members.insert(self.ref_slot % 6, self.names[self.refs[i]])
        return (self.names[self.refs[i]], self.names[self.targets[i]], str(self.captions[i]), members,
                self.K_sorted_index_names[i].tolist(), self.K_labels[i], self.K_group_labels[i])


def pair_keep(seed: int, rows: int, cols: int, p: float) -> torch.Tensor:
    """The dropout mask of the FUSED training kernels regenerated on the host (include/cirrank.h, cir_residual_layernorm_train): element
    (row, col) is kept iff the 16 bits of its column pair's hash32(row_key ^ (col >> 1)) - low half for even, high half for odd col - are
    >= round(p * 65536)."""
    m32 = np.uint64(0xFFFFFFFF)
    row = np.arange(rows, dtype=np.uint64)
    with np.errstate(over="ignore"):
        rk = (np.uint64(seed & 0xFFFFFFFF) + np.uint64(seed >> 32) * np.uint64(0x85EBCA6B) + (row & m32) * np.uint64(0x9E3779B9)
              + (row >> np.uint64(32)) * np.uint64(0xC2B2AE35)) & m32
        x = rk[:, None] ^ (np.arange(cols, dtype=np.uint64)[None, :] >> np.uint64(1))
        x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & m32
        x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & m32
        x ^= x >> np.uint64(16)
    odd = (np.arange(cols) & 1).astype(bool)[None, :]
    u16 = np.where(odd, x >> np.uint64(16), x & np.uint64(0xFFFF))
    thr = int(np.float32(p) * np.float32(65536.0) + np.float32(0.5))
    return torch.from_numpy(u16 >= np.uint64(thr))


def splitmix_keep(seed: int, n: int, p: float) -> torch.Tensor:
    """The dropout mask of the STAND-ALONE operators (cir_eltwise mode 4, cir_softmax_dropout; train.hip uniform01): element i of a launch is
    kept iff the top 24 bits of splitmix64(seed, i), as a fraction, are >= p."""
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (idx + np.uint64(1))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return torch.from_numpy(u >= np.float32(p))


def dropout_hooks(tr, bsz: int, l: int, n: int, d: int, heads: int, p: float):
    """(drop, keep) for `oracle.cir_oracle.img_txt_fusion_train(drop=)`: the masks of every dropout site of the LAST forward of `tr`
    (a train.NlvrTrainer), regenerated on the host from the kernels' counters - the seeds of `tr._site`, the element numbering of each site
    (candidate-major triplets t = j * B + i, both branches stacked in the FFN and in the attention launches' groups)."""
    r = bsz * bsz * l
    cache = {}

    def keep(kind, layer, b):
        key = (kind, layer, b)
        if key not in cache:
            if kind == "emb":
                cache[key] = splitmix_keep(tr._site(9000), r * d, p).view(bsz, bsz, l, d)                        # [target j][query i]; cir_eltwise's generator
            elif kind == "self_out":
                cache[key] = pair_keep(tr._site(layer, b, 2), r, d, p).view(bsz, bsz, l, d)
            elif kind == "cross_out":
                cache[key] = pair_keep(tr._site(layer, 2, 4), r, d, p).view(bsz, bsz, l, d)
            elif kind == "ffn_out":                                                                            # both branches stacked: 2R rows
                cache[key] = pair_keep(tr._site(layer, 0, 5), 2 * r, d, p).view(2, bsz, bsz, l, d)
            elif kind == "self_attn":                                                      # ONE site for both branches: group = (branch, triplet j * B + i)
                cache[key] = pair_keep(tr._site(layer, 0, 1), 2 * bsz * bsz * heads * l, l, p).view(2, bsz, bsz, heads, l, l)[b]
            elif kind == "cross_attn":                                                     # group = (branch, target j), rows = (query i, token)
                cache[key] = pair_keep(tr._site(layer, 0, 3), 2 * bsz * heads * bsz * l, n, p).view(2, bsz, heads, bsz, l, n)[b]
        return cache[key]

    def drop(kind, layer, b, qi, x):
        k = keep(kind, layer, b)
        if kind == "ffn_out":
            mk = k[b][:, qi]
        elif kind == "cross_attn":
            mk = k[:, :, qi]
        else:
            mk = k[:, qi]
        assert mk.shape == x.shape, (kind, mk.shape, x.shape)
        return x * mk.to(x.dtype) / (1.0 - p)

    return drop, keep
