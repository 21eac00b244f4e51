"""Runs the product HIP sources on the CPU under tests/hipemu (fibers + emulated wave ops) and
checks them against oracle/.  This is a *debugging aid for the kernels' index logic* — it is not a
product path and proves nothing about the GPU build; the real parity tests are tests/test_engine_gpu.py."""
import os

import numpy as np
import pytest

import engine_checks as ec
from microwakeword_amd import native

@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "data_golden.npz"))


@pytest.mark.parametrize("tag", ["u16", "f32"])
def test_get_data_reference_golden(emu_lib, gold, tag):
    ec.check_get_data_against_reference_golden(emu_lib, gold, tag)


def test_sampler_descriptors(emu_lib):
    ec.check_sampler_matches_oracle_descriptors(emu_lib, B=48, n_samples=40)


@pytest.mark.parametrize("training", [False, True])
def test_forward(emu_lib, training):
    ec.check_forward_parity(emu_lib, B=3, T=194, training=training, grid=2)


def test_forward_short_and_ragged_tiles(emu_lib):
    ec.check_forward_parity(emu_lib, B=2, T=111, training=True, grid=1)
    ec.check_forward_parity(emu_lib, B=1, T=60, training=False)


def test_train_steps(emu_lib):
    ec.check_train_steps(emu_lib, B=5, T=194, steps=2, grid=2)


def test_saturated_logits_loss_forms(emu_lib):
    ec.check_saturated_logits_loss(emu_lib, B=4, T=60)


def test_variable_batch_sizes_do_not_leave_stale_statistics(emu_lib):
    ec.check_variable_batch_sizes(emu_lib, T=60, sizes=(10, 3, 3, 1, 10))
    ec.check_variable_batch_sizes(emu_lib, T=60, sizes=(9, 2, 2), graphs=True)


def test_gradients_without_imposed_relu_masks(emu_lib):
    ec.check_gradients_unimposed(emu_lib, B=6, T=130, bound=1e-2)


def test_train_steps_other_lengths(emu_lib):
    ec.check_train_steps(emu_lib, B=3, T=130, steps=1, grid=4)
    ec.check_train_steps(emu_lib, B=2, T=60, steps=1, grid=1, graphs=True)


def test_notebook_topology_stride3_mixconv_groups(emu_lib):
    """first conv 5x1 stride 3, 64 filters, MixConv [7,11] / [9,15] groups (fused with zero taps + gradient mask)."""
    ec.check_forward_parity(emu_lib, B=2, T=204, training=True, grid=2, flags=ec.NOTEBOOK)
    ec.check_train_steps(emu_lib, B=3, T=204, steps=1, grid=2, flags=ec.NOTEBOOK)
    # grids above this topology's defaults (2 forward / 1 backward workgroup per CU): the partial rows are sized for the options' maxima
    ec.check_train_steps(emu_lib, B=9, T=231, steps=1, grid=8, flags=ec.NOTEBOOK)


@pytest.mark.parametrize("training", [False, True])
def test_inception_forward(emu_lib, training):
    ec.check_inception_forward(emu_lib, B=3, T=194, training=training, grid=2)


def test_inception_train_steps(emu_lib):
    ec.check_inception_train_steps(emu_lib, B=4, T=194, steps=2, grid=2)


def test_inception_variant_dilation_groups_two_stems(emu_lib):
    ec.check_inception_forward(emu_lib, B=2, T=120, training=True, grid=1, flags=ec.INC_VARIANT)
    ec.check_inception_train_steps(emu_lib, B=3, T=120, steps=1, grid=2, graphs=True, flags=ec.INC_VARIANT)


def test_inception_statistics_hand_over_matches_finalize_launches(emu_lib):
    ec.check_inception_bn_inline_matches_finalize(emu_lib, B=5, T=120, steps=3)
    ec.check_inception_bn_inline_matches_finalize(emu_lib, B=4, T=120, steps=2, flags=ec.INC_VARIANT)


def test_frame_chunks_of_the_pointwise_graph_ops(emu_lib):
    """"graph_frame_chunks": the 1x1 ops of a conv/BN graph process a window as 2-4 frame chunks (smaller LDS tiles); same
    results against the oracle as with whole windows, uneven last chunks included."""
    ec.check_inception_train_steps(emu_lib, B=3, T=121, steps=2, grid=2, options={"graph_frame_chunks": 2})
    ec.check_inception_train_steps(emu_lib, B=3, T=120, steps=1, grid=0, options={"graph_frame_chunks": 3})
    ec.check_inception_train_steps(emu_lib, B=2, T=194, steps=1, grid=0, options={"graph_frame_chunks": 1})   # automatic: only the 24 -> 30, 10 -> 48 and 48 -> 16 ops are chunked
    ec.check_inception_train_steps(emu_lib, B=3, T=120, steps=1, grid=2, flags=ec.INC_VARIANT, options={"graph_frame_chunks": 4})
    ec.check_graph_mixednet(emu_lib, ec.GRAPH_MIXEDNET, B=3, T=100, steps=2, grid=0, options={"graph_frame_chunks": 3})
    ec.check_graph_mixednet(emu_lib, ec.GRAPH_MIXEDNET_NOCONV1, B=2, T=60, steps=1, grid=1, graphs=True, options={"graph_frame_chunks": 2})


def test_graph_grid_options(emu_lib):
    ec.check_graph_grid_options(emu_lib, B=5, T=100)


def test_inception_generated_dropout(emu_lib):
    ec.check_inception_generated_dropout(emu_lib, B=3, T=120)


def test_validation_on_device(emu_lib, gold):
    ec.check_validation_on_device(emu_lib, gold, "u16")


@pytest.mark.parametrize("which", range(5))
def test_crossed_topologies_on_the_block_kernels(emu_lib, which):
    """Either documented width with either kernel set and either first conv: all on the specialised block kernels."""
    flags = ec.CROSSED[which]
    ec.check_train_steps(emu_lib, B=3, T=204 if flags.get("stride", 1) == 3 else 150, steps=1, grid=2, flags=flags)


def test_bf16_pointwise_mode(emu_lib):
    """BASELINE configs[4]: 1x1 contractions with bf16 operands, against the oracle rounding the same operands."""
    ec.check_forward_parity(emu_lib, B=2, T=111, training=True, grid=2, flags=ec.BF16)
    ec.check_train_steps(emu_lib, B=3, T=130, steps=1, grid=2, flags=ec.BF16)


def test_bf16_storage_mode(emu_lib):
    """... with p_k / g_k held as bf16 in HBM ("storage_bf16"), against the oracle rounding the same stored tensors
    (statistics from the unrounded values, see oracle/model_oracle.py _StoredBatchNorm)."""
    ec.check_forward_parity(emu_lib, B=2, T=111, training=True, grid=2, flags=ec.BF16_STORED)
    ec.check_forward_parity(emu_lib, B=2, T=111, training=False, grid=1, flags=ec.BF16_STORED)
    ec.check_train_steps(emu_lib, B=3, T=130, steps=1, grid=2, flags=ec.BF16_STORED)


def test_inception_unfused_branch_heads(emu_lib):
    """The 22-op form (one op per Keras layer) stays available and agrees as well."""
    ec.check_inception_forward(emu_lib, B=2, T=194, training=True, grid=2, fuse_heads=False)
    ec.check_inception_train_steps(emu_lib, B=3, T=194, steps=1, grid=2, fuse_heads=False)


def test_train_loop_end_to_end(emu_lib, tmp_path):
    ec.check_train_loop_end_to_end(emu_lib, tmp_path, B=8, steps=9)


def test_mixednet_on_generic_graph_kernels(emu_lib):
    """Flag combinations outside the specialised kernels (repeat 2, a block without depthwise, odd filters,
    strided 5x1 first conv; and no first conv at all) run on the conv/BN graph kernels."""
    ec.check_graph_mixednet(emu_lib, ec.GRAPH_MIXEDNET, B=3, T=100, steps=1, grid=2)
    ec.check_graph_mixednet(emu_lib, ec.GRAPH_MIXEDNET, B=3, T=100, steps=3, grid=0, graphs=True)   # per-launch grids, both accumulator parities, replayed
    ec.check_graph_mixednet(emu_lib, ec.GRAPH_MIXEDNET, B=3, T=100, steps=1, grid=2, bn_inline=0)   # finalize launches
    ec.check_graph_mixednet(emu_lib, ec.GRAPH_MIXEDNET_NOCONV1, B=2, T=60, steps=1, grid=1, graphs=True)


def test_mixednet_residual_connections(emu_lib):
    """residual_connection (1x1 conv + BN of the block input, StridedDrop, added before every repeat's ReLU,
    including the last block that feeds the classifier head) on the graph kernels."""
    ec.check_graph_mixednet(emu_lib, ec.GRAPH_MIXEDNET_RESIDUAL, B=3, T=80, steps=1, grid=2)


def test_mixednet_attention_and_pooled_heads(emu_lib):
    """spatial_attention / pooled / max_pool heads (mixednet.py:234-275,362-381), forward and the full backward."""
    for flags in ec.GRAPH_MIXEDNET_HEADS:
        ec.check_graph_mixednet(emu_lib, flags, B=3, T=64, steps=1, grid=2)


def test_mixednet_model_selects_kernels_by_shape(emu_lib):
    """mixednet.model(): specialised block kernels when the shape is instantiated, generic graph kernels
    otherwise, NotImplementedError for the options nothing implements."""
    from microwakeword_amd import mixednet
    m = mixednet.model(ec.DEF, (194, 40), 4, lib=emu_lib, max_batch=4)
    assert m.name == "mixednet" and m.engine.n_params == 22177
    m.engine.close()
    g = mixednet.model(ec.GRAPH_MIXEDNET, (100, 40), 4, lib=emu_lib, max_batch=4)
    assert "generic" in g.name
    x = ec.synth_x(np.random.default_rng(0), 3, 100)
    assert g.predict_on_batch(x).shape == (3, 1)
    lines = []
    g.summary(print_fn=lines.append)
    assert any("depthwise" in ln for ln in lines)
    g.engine.close()
    r = mixednet.model(dict(ec.DEF, residual_connection="0,1,0,0"), (194, 40), 4, lib=emu_lib, max_batch=4)
    assert "generic" in r.name
    r.engine.close()
    a = mixednet.model(dict(ec.DEF, spatial_attention=1, pooled=1), (194, 40), 4, lib=emu_lib, max_batch=4)
    assert "generic" in a.name and a.layout.t_last == 1
    a.engine.close()


def test_smallest_spectrogram(emu_lib):
    """T = 47: exactly one output frame after the last block."""
    ec.check_forward_parity(emu_lib, B=2, T=47, training=True, grid=1)
    ec.check_train_steps(emu_lib, B=2, T=47, steps=1, grid=1)


def test_cli_end_to_end_from_disk(emu_lib, tmp_path, monkeypatch):
    """`python -m microwakeword_amd.model_train_eval --training_config cfg.yaml mixednet ...` (model_train_eval.py:
    277-439): YAML config, `<features_dir>/<mode>/*_mmap` stores on disk, shape derivation, model summary, train
    loop with validation and checkpoints — and a second invocation that restores the checkpoint."""
    import yaml

    from microwakeword_amd import model_train_eval, ragged
    rng = np.random.default_rng(0)
    for prov, positive in (("wake", True), ("background", False)):
        for mode, n in (("training", 12), ("validation", 6), ("validation_ambient", 2)):
            if mode == "validation_ambient" and positive:
                continue
            lo, hi = (200, 260) if mode == "validation_ambient" else (62, 90)
            samples = []
            for _ in range(n):
                s = rng.integers(0, 200, size=(int(rng.integers(lo, hi)), 40)).astype(np.uint16)
                if positive:
                    s[-30:-10, 8:24] += 400
                samples.append(s)
            ragged.write_ragged_store(str(tmp_path / prov / mode / ("%s_mmap" % mode)), samples)
    cfg = dict(window_step_ms=10, train_dir=str(tmp_path / "trained"), clip_duration_ms=160, batch_size=4, training_steps=[4],
               learning_rates=[0.001], eval_step_interval=2, target_minimization=0.9, minimization_metric=None,
               maximization_metric="average_viable_recall", time_mask_max_size=[3], time_mask_count=[1], freq_mask_max_size=[3],
               freq_mask_count=[1], positive_class_weight=[1], negative_class_weight=[2],
               features=[dict(features_dir=str(tmp_path / "wake"), sampling_weight=1.0, penalty_weight=1.0, truth=True,
                              truncation_strategy="truncate_start", type="mmap"),
                         dict(features_dir=str(tmp_path / "background"), sampling_weight=2.0, penalty_weight=1.0, truth=False,
                              truncation_strategy="random", type="mmap")])
    (tmp_path / "cfg.yaml").write_text(yaml.dump(cfg))
    monkeypatch.setenv("MWW_HIP_LIB", emu_lib.path)   # the CLI builds its own engine: point it at the emulator build
    argv = ["--training_config", str(tmp_path / "cfg.yaml"), "--verbosity", "ERROR", "mixednet", "--residual_connection", "0,0,0,0"]
    out = model_train_eval.main(argv)
    assert set(out) == {"best_minimization", "best_maximization", "best_no_faph_cutoff"}
    run = tmp_path / "trained"
    for f in ("training_config.yaml", "model_summary.txt", "best_weights.weights.h5.npz", "last_weights.weights.h5.npz",
              "restore/ckpt.weights.npz", "restore/ckpt.opt.npz", "logs/train/scalars.jsonl", "logs/validation/scalars.jsonl"):
        assert (run / f).exists(), f
    saved = yaml.load((run / "training_config.yaml").read_text(), yaml.Loader)   # the reference's dump carries a python tuple
    assert saved["spectrogram_length"] == 60 and saved["spectrogram_length_final_layer"] == 14
    from microwakeword_amd.layout import MixedNetLayout
    total = MixedNetLayout(ec.DEF, 60).keras_param_counts()[0]
    assert total == 22561 - 48 * (148 - 14) and "Total params: %d" % total in (run / "model_summary.txt").read_text()
    with pytest.raises(ValueError, match="already exists"):
        model_train_eval.main(argv)                                  # model_train_eval.py:118-120
    out2 = model_train_eval.main(argv[:2] + ["--restore_checkpoint", "1"] + argv[2:])
    assert set(out2) == set(out)
    z = np.load(run / "restore" / "ckpt.opt.npz")
    assert int(z["step"]) == 8                                        # 4 restored + 4 new optimizer steps


def test_data_path_fuzz(emu_lib):
    ec.check_data_fuzz(emu_lib, cases=10)


def test_mixednet_topology_fuzz(emu_lib):
    ec.check_topology_fuzz(emu_lib, cases=5)


def test_against_frozen_oracle_outputs(emu_lib, golden_dir):
    ec.check_against_frozen_oracle(emu_lib, golden_dir)


def test_against_the_reference_graph_fixture(emu_lib, golden_dir):
    """tests/golden/ref_graph_golden.npz = the reference's own mixednet.py / inception.py executed over float64 stand-ins of the Keras
    layer primitives (oracle/ref_model_shim.py): probabilities, loss, every gradient, BN moving statistics."""
    print(ec.check_against_reference_graph_fixture(emu_lib, golden_dir))


def test_prefetched_batches_train_like_the_synchronous_sampler(emu_lib):
    ec.check_prefetched_training_matches_synchronous(emu_lib)


def test_train_loop_prefetch_is_schedule_only(emu_lib, tmp_path):
    ec.check_train_loop_prefetch_is_schedule_only(emu_lib, tmp_path, B=6, steps=5)


def test_train_loop_data_parallel_world1(emu_lib, tmp_path):
    dp = ec.check_train_loop_data_parallel_world1(emu_lib, tmp_path, "gloo")
    assert not dp.library_comm and dp.engine_driven


def test_first_conv_tail_rows(emu_lib):
    ec.check_first_conv_tail_rows(emu_lib, B=2, lengths=(203, 206), topologies=(ec.NOTEBOOK,))


@pytest.mark.parametrize("wide", [0, 1])
def test_first_block_tail_k_step_with_a_three_tap_depthwise(emu_lib, wide):
    """A 3-tap first depthwise behind a strided first convolution has TAIL = 2 tail rows, but the tail k-step of the
    dW1 contraction always reads four (round-5 advisor finding: g0 / x rows past the staged ones).  Lengths in tail mode
    (Ta in {65, 66}), both forms of the first-block backward."""
    for stride, lengths in ((3, (197, 200)), (2, (132, 134))):
        flags = dict(ec.DEF, mixconv_kernel_sizes="[3],[9],[13],[21]", stride=stride, bwd_wide=wide)
        for T in lengths:
            ec.check_train_steps(emu_lib, B=3, T=T, steps=1, grid=2, flags=flags)
    # ... and a 7-tap first depthwise has TAIL = 6 tail rows: TWO tail k-steps (round-6 fuzz finding, tools/gpu_table_fuzz.py case
    # 556: with one, rows TT + 4 and TT + 5 of a 69- / 70-row window never reached the conv1 weight gradient: error 1e-2)
    for stride, T in ((3, 212), (2, 141)):
        ec.check_train_steps(emu_lib, B=3, T=T, steps=1, grid=2, flags=dict(ec.DEF, mixconv_kernel_sizes="[7],[9],[13],[21]", stride=stride, bwd_wide=wide))


def test_conv1_x6_against_the_exact_fp32_form(emu_lib):
    """The first convolution and its weight gradient as bf16 slice products (the default for stride-1 first convolutions) against
    the exact-fp32 MFMA form of the same kernels, incl. values over the whole uint16 range (three-slice x)."""
    ec.check_conv1_x6_against_the_f32_form(emu_lib, B=5, T=150)
    ec.check_conv1_x6_against_the_f32_form(emu_lib, B=3, T=111, raw_u16_range=True)
    ec.check_conv1_x6_against_the_f32_form(emu_lib, B=3, T=150, flags=dict(ec.DEF, first_conv_kernel_size=5, pointwise_filters="32,48,64,48"))


def test_thirty_two_channel_head_beyond_256_frames(emu_lib):
    """A 32-channel last block with more than 256 final frames (round-5 advisor finding: shape_supported accepted it, launch_head had
    no instantiation: 'no head kernel for this (channels, frames) shape' at the first train step)."""
    flags = dict(ec.DEF, pointwise_filters="48,48,48,32")
    from microwakeword_amd import mixednet
    assert mixednet.kernel_family(flags, 330, lib=emu_lib)[0] == "block"
    ec.check_forward_parity(emu_lib, B=2, T=330, training=True, grid=2, flags=flags)
    ec.check_train_steps(emu_lib, B=2, T=330, steps=1, grid=2, flags=flags)


def test_head_frames_beyond_the_widest_instantiation_are_refused_at_creation(emu_lib):
    """round-6 fuzz finding (tools/gpu_x6_fuzz.py case 460): 64 channels x 390 final frames passed shape_supported and failed in
    launch_head at the first forward; now MWW_ERR_UNSUPPORTED when the model is created."""
    ec.check_head_frame_limit_is_refused_at_creation(emu_lib)


def test_wide_first_block_backward_with_x6(emu_lib):
    """Option "bwd_first_wide": the 512-thread form of the stride-1 first block's backward kernel with the conv1 weight gradient as
    bf16 slice products (kernels_bwdw.hip.h bwd_firstw_kernel<..., X6>), against the oracle: ragged tiles, several windows per
    workgroup, 32 / 48 / 64 pointwise filters (the g0 planes reach into the u tile's space at 32)."""
    ec.check_train_steps(emu_lib, B=5, T=194, steps=1, grid=2, flags=dict(ec.DEF, bwd_first_wide=1))
    for flags in (dict(ec.DEF, pointwise_filters="32,48,48,48"), dict(ec.DEF, pointwise_filters="64,64,64,64", mixconv_kernel_sizes="[7],[9],[13],[21]")):
        ec.check_train_steps(emu_lib, B=3, T=130, steps=1, grid=4, flags=dict(flags, bwd_first_wide=1))
    ec.check_gradients_unimposed(emu_lib, B=6, T=130, bound=1e-2, flags=dict(ec.DEF, bwd_first_wide=1))


def test_inception_static_shapes_are_schedule_only(emu_lib):
    ec.check_inception_static_shapes_are_schedule_only(emu_lib, B=4, lengths=(100, 236), steps=2, grid=2, combos=((0, 0, 0), (1, 1, 0), (1, 1, 1)))


def test_bn_inline_matches_finalize(emu_lib):
    ec.check_bn_inline_matches_finalize(emu_lib, B=5, T=100, steps=3)


@pytest.mark.parametrize("dtype", ["u16", "f32"])
def test_fused_input_is_bit_identical(emu_lib, dtype):
    ec.check_fused_input(emu_lib, dtype=dtype)


def test_fused_input_through_captured_graphs(emu_lib):
    ec.check_fused_input(emu_lib, B=4, steps=9, graphs=True)   # more steps than mailbox slots (8): a slot's graph is replayed


def test_shape_table_fuzz(emu_lib):
    """Random flag sets out of the block kernels' shape table with random (frames, batch, grid) - tools/gpu_table_fuzz.py, which runs
    thousands of them on the GPU - here on the emulator, whose LDS starts as NaN and whose allocations end at a guard page: an
    uninitialised LDS read or an access past a buffer shows up here and not on the device."""
    import sys
    from microwakeword_amd import mixednet
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    sys.path.insert(0, tools)
    try:
        from gpu_table_fuzz import random_table_flags
    finally:
        sys.path.remove(tools)
    for case in range(5000, 5016):
        flags, T, B, grid = random_table_flags(case)
        assert mixednet.kernel_family(flags, T, lib=emu_lib)[0] == "block", (case, flags)
        try:
            ec.check_train_steps(emu_lib, B=min(B, 6), T=T, steps=1, grid=grid, flags=flags)   # (the float64 oracle is what takes the time)
        except AssertionError as e:
            raise AssertionError("case %d %s T %d B %d grid %d: %s" % (case, flags, T, B, grid, e))


def test_gather_fuzz(emu_lib):
    ec.check_gather_fuzz(emu_lib, cases=1, first=11)   # the short-window cases of the fuzz (the GPU suite runs 46)
    ec.check_gather_fuzz(emu_lib, cases=1, first=5)


def test_inception_topology_fuzz(emu_lib):
    ec.check_inception_topology_fuzz(emu_lib, cases=1, first=76, B=2, T=70)   # unfused heads with sub-spectral groups


def test_inception_stem_gathers_descriptor_only_batches(emu_lib):
    # (more cases, window lengths, captured graphs and grids: tools/gpu_inc_fuzz.py / test_engine_gpu.py on the device)
    ec.check_inception_gathered_stem(emu_lib, cases=1, B=4, lengths=(194,), rounds=1)
    ec.check_inception_gathered_stem(emu_lib, cases=1, first=4, B=6, grid=2, graphs=(1,), lengths=(150,), rounds=2)



@pytest.mark.parametrize("wide", [0, 1])
def test_both_forms_of_the_block_backward_kernels(emu_lib, wide):
    """bwd_blockw_kernel (kernels_bwdw.hip.h, 512 threads per 64-row tile: the default) and bwd_block_kernel (256)."""
    flags = dict(ec.DEF, bwd_wide=wide)
    ec.check_train_steps(emu_lib, B=5, T=194, steps=2, grid=2, flags=flags)
    ec.check_train_steps(emu_lib, B=3, T=130, steps=1, grid=4, flags=flags)
    ec.check_gradients_unimposed(emu_lib, B=6, T=130, bound=1e-2, flags=flags)
    ec.check_train_steps(emu_lib, B=2, T=60, steps=1, grid=1, graphs=True, flags=flags)


def test_wide_block_backward_kernels_notebook_and_crosses(emu_lib):
    ec.check_train_steps(emu_lib, B=3, T=204, steps=1, grid=2, flags=dict(ec.NOTEBOOK, bwd_wide=1))
    for flags in ec.CROSSED[:3]:
        ec.check_train_steps(emu_lib, B=3, T=204 if flags.get("stride", 1) == 3 else 150, steps=1, grid=2, flags=dict(flags, bwd_wide=1))


def test_block_kernels_of_the_wider_shape_table(emu_lib):
    """Shapes the round-5 table added (csrc/block_launch.hip.h): 32-wide and mixed-width blocks, kernel lengths 3 / 7 / 17 / 19,
    a stride-2 first convolution, five blocks - all on the specialised block kernels (the summary says which family)."""
    from microwakeword_amd import mixednet
    cases = [dict(ec.DEF, pointwise_filters="32,32,32,32", mixconv_kernel_sizes="[3],[7],[17],[19]"),
             dict(ec.DEF, pointwise_filters="32,48,64,48", mixconv_kernel_sizes="[7],[5,9],[3],[11,15]", first_conv_kernel_size=5, stride=2),
             dict(ec.DEF, pointwise_filters="64,32,32,48,48", mixconv_kernel_sizes="[3],[5],[7],[9],[11]", repeat_in_block="1,1,1,1,1",
                  residual_connection="0,0,0,0,0")]
    for flags in cases:
        assert mixednet.kernel_family(flags, 150, lib=emu_lib)[0] == "block", flags
        ec.check_forward_parity(emu_lib, B=2, T=150, training=True, grid=2, flags=flags)
        ec.check_train_steps(emu_lib, B=3, T=150, steps=1, grid=2, flags=flags)
    m = mixednet.model(cases[0], (150, 40), 2, lib=emu_lib, max_batch=2)
    lines = []
    m.summary(print_fn=lines.append)
    assert any(l.startswith("Kernels: specialised block kernels") for l in lines)
