"""Registry of the launch-coverage guard (tests/test_gpu_zz_coverage.py); one module instance shared by conftest.py and the guard."""
# ---------------------------------------------------------------------------------------------------------------------------------
# Launch coverage (tests/test_gpu_zz_coverage.py): the kernel instantiations launched INSIDE tests that compare the HIP path with the
# CPU oracle / an fp64 reference of the same operator / a reference-generated golden fixture are collected here, by test function
# name.  HIP-vs-HIP tests (fused-vs-unfused, linearity, reproducibility, data-parallel equivalence) are deliberately NOT listed.
ORACLE_COMPARED = {
    # tests/test_gpu_kernels.py -- one operator through the C ABI vs fp64 / fp32 PyTorch of the same operator
    'test_gemm_forward_nt', 'test_gemm_catches_transposes', 'test_gemm_dgrad_nn', 'test_gemm_wgrad_tn_with_bias_grad',
    'test_gemm_pair_dgrad_and_wgrad_tight', 'test_gemm_pair_with_a_second_wgrad_riding_on_the_launch', 'test_gemm_column_sums_for_the_following_batchnorm', 'test_gemm_rowstream_convolution', 'test_gemm_wgrad_into_a_sub_matrix', 'test_token_gradients_default_and_deterministic',
    'test_gemm_dgrad_splitk_planes_and_their_sum_in_layernorm_bwd', 'test_gemm_wgrad_group_full_k_deterministic',
    'test_fused_layernorm_backward_chain_row_statistics', 'test_dgrad_with_whole_row_layernorm_backward',
    'test_gemm_epilogues_gelu_resid_token_dgelu', 'test_gemm_fat_forward_tile', 'test_gemm_fat_dgrad_tile', 'test_layernorm_fwd_bwd', 'test_attention_fwd_bwd',
    'test_attention_weight_dropout_uses_the_oracle_mask', 'test_attention_single_plane_probabilities_keep_the_row_statistics_and_stay_within_bf16_of_the_full_split', 'test_attention_block_diagonal_segments', 'test_tokenizer_modules_match_oracle',
    'test_head_and_cross_entropy', 'test_adam_matches_torch_and_refreshes_planes', 'test_block_fwd_bwd_matches_oracle',
    'test_group_encoder_layer_fwd_bwd_matches_oracle', 'test_assemble_tokens_fwd_bwd',
    # tests/test_gpu_model.py -- the engine vs reference goldens / the oracle's autograd
    'test_engine_matches_reference_golden', 'test_split_precision_backward_matches_reference_gradients_tightly',
    'test_drop_in_module_training_step_matches_oracle', 'test_fused_train_step_matches_oracle_and_graph_replay',
    'test_cfg2_full_size_parity', 'test_group_embed_training_mode_dropout_matches_oracle',
    'test_forward_images_and_lwf_gradients_match_reference_golden', 'test_drop_in_module_lwf_step_and_frozen_stem',
    'test_lwf_train_step_matches_oracle_adam_step', 'test_cfg3_reduced_batch_training_step_matches_oracle',
    # tests/test_gpu_points.py / test_gpu_fullsize.py
    'test_fps_indices_bit_exact', 'test_knn16_and_3nn_indices_bit_exact', 'test_batchnorm_relu_max_fwd_bwd', 'test_gather_scatter_interp',
    'test_point_engine_matches_reference_golden', 'test_point_engine_variants_match_reference_golden',
    'test_drop_in_point_module_matches_reference', 'test_drop_in_variant_module_lwf_step_matches_reference',
    'test_drop_in_variant_modules_points_only', 'test_group_project_fwd_bwd', 'test_mean_points_and_broadcast',
    'test_cfg4_full_size_parity', 'test_cfg5_full_size_parity', 'test_cfg3_forward_at_batch_13_reaches_the_fat_residual_tile_and_matches_oracle', 'test_am_softmax_row_head_kernels', 'test_sgd_momentum_matches_torch_and_refreshes_planes',
    'test_drop_in_seg_module_with_am_softmax_head_and_cls_error', 'test_point_trained_state_fixture_from_the_reference_sgd',
    # tests/test_gpu_trajectory.py, tests/test_eval_binvox.py
    'test_trajectory_tracks_the_oracle', 'test_benched_backward_reproduces_the_reference_trained_accuracy_and_final_loss', 'test_split_backward_tracks_the_reference_seed_by_seed_on_the_stable_fixture', 'test_trained_state_fixture_from_the_reference_optimizer', 'test_trained_state_fixture_default_backward_trains_alike', 'test_gpu_unpack_voxels', 'test_gpu_cls_eval_matches_reference_fixture',
    'test_gpu_partseg_eval_matches_reference_fixture', 'test_gpu_partseg_eval_large_random_vs_oracle',
}
COVERED = {}          # 'family:key' -> the oracle-compared tests that launched it


def collect(lib):
    import ctypes
    n = lib.s3d_cov_collect(None, 0)
    buf = ctypes.create_string_buffer(int(n) + 8)
    lib.s3d_cov_collect(buf, int(n) + 8)
    out = {}
    for line in buf.value.decode().splitlines():
        fam, key, cnt = line.rsplit(':', 2)
        out[f'{fam}:{key}'] = int(cnt)
    return out


