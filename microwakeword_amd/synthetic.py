"""Synthetic feature stores of the benchmark configuration: two providers
(label 1 / 0, sampling_weight 2 / 10, penalty 1 / 1, ``truncate_start`` / ``random``), 4096 ragged
samples each, lengths U{150..400} frames, raw micro-frontend uint16 values U{0..666}
(or an already-scaled float32 variant), ``numpy.random.default_rng(1234)``."""
import numpy as np


def synthetic_stores(n_samples=4096, seed=1234, dtype=np.uint16, min_len=150, max_len=400):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(2):
        lens = rng.integers(min_len, max_len + 1, size=n_samples)
        if dtype == np.uint16:
            st = [rng.integers(0, 667, size=(int(l), 40), dtype=np.uint16) for l in lens]
        else:
            st = [(rng.random((int(l), 40), dtype=np.float32) * np.float32(26.0)) for l in lens]
        out.append(st)
    return out


def benchmark_config(n_samples=4096, seed=1234, dtype=np.uint16, n_val=0, n_ambient=0):
    """A ``config`` dict for :class:`microwakeword_amd.data.FeatureHandler` holding the stores in RAM.
    ``n_val`` adds that many validation samples to each provider, ``n_ambient`` long negative
    "validation_ambient" recordings (600..1500 frames) that the split strategy cuts into windows."""
    pos, neg = synthetic_stores(n_samples, seed, dtype)
    rng = np.random.default_rng(seed + 1)

    def extra(n, lo, hi):
        return [rng.integers(0, 667, size=(int(l), 40), dtype=np.uint16) for l in rng.integers(lo, hi + 1, size=n)]

    pos_modes, neg_modes = {"training": [pos]}, {"training": [neg]}
    if n_val:
        pos_modes["validation"] = [extra(n_val, 150, 400)]
        neg_modes["validation"] = [extra(n_val, 150, 400)]
    if n_ambient:
        neg_modes["validation_ambient"] = [extra(n_ambient, 600, 1500)]
    return {"stride": 1, "window_step_ms": 10, "features": [
        dict(type="mmap", stores=pos_modes, truth=True, sampling_weight=2.0, penalty_weight=1.0,
             truncation_strategy="truncate_start"),
        dict(type="mmap", stores=neg_modes, truth=False, sampling_weight=10.0, penalty_weight=1.0,
             truncation_strategy="random")]}, (pos, neg)


DEFAULT_MIXEDNET_FLAGS = dict(pointwise_filters="48, 48, 48, 48", residual_connection="0,0,0,0", repeat_in_block="1,1,1,1",
                              mixconv_kernel_sizes="[5], [9], [13], [21]", max_pool=0, first_conv_filters=32,
                              first_conv_kernel_size=3, spatial_attention=0, pooled=0, stride=1)
# the MixedNet flags of the reference's training notebook (cell 10); spectrogram_length 204
NOTEBOOK_MIXEDNET_FLAGS = dict(DEFAULT_MIXEDNET_FLAGS, first_conv_kernel_size=5, stride=3, first_conv_filters=32,
                               pointwise_filters="64,64,64,64", mixconv_kernel_sizes="[5],[7,11],[9,15],[23]")
DEFAULT_INCEPTION_FLAGS = dict(cnn1_filters="24", cnn1_kernel_sizes="5", cnn1_subspectral_groups="4", cnn2_filters1="10,10,16",
                               cnn2_filters2="10,10,16", cnn2_kernel_sizes="5,5,5", cnn2_subspectral_groups="1,1,1",
                               cnn2_dilation="1,1,1", dropout=0.2)
SPEC_AUGMENT_POLICY = dict(freq_mix_prob=0.0, time_mask_max_size=5, time_mask_count=2, freq_mask_max_size=5, freq_mask_count=2)
