"""ctypes door onto libcrfconv_amd.so (the C ABI declared in include/crfconv_amd.h).

The library is the product: there is no Python / PyTorch fallback.  If it is missing or a call
fails, an exception is raised -- never a silent slow path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CRFCONV_LIB') or os.path.join(_HERE, 'libcrfconv_amd.so')      # CRFCONV_LIB: A/B builds of scratch/

_vp, _i, _i64, _sz, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_float
_d = ctypes.c_double
_u64 = ctypes.c_uint64

# name -> (restype, argtypes); mirrors include/crfconv_amd.h one to one
SIGNATURES = {
    'crfconv_abi_version': (_i, []),
    'crfconv_last_error': (ctypes.c_char_p, []),
    'crfconv_knn': (_i, [_vp, _sz, _sz, _vp, _sz, _sz, _vp]),
    'crfconv_knn_omp': (_i, [_vp, _sz, _sz, _vp, _sz, _sz, _vp]),
    'crfconv_knn_batch': (_i, [_vp, _sz, _sz, _sz, _vp, _sz, _sz, _vp]),
    'crfconv_knn_batch_omp': (_i, [_vp, _sz, _sz, _sz, _vp, _sz, _sz, _vp]),
    'crfconv_knn_batch_dev_workspace': (_sz, [_sz, _sz, _sz, _sz]),
    'crfconv_knn_batch_dev': (_i, [_vp, _sz, _sz, _sz, _vp, _sz, _sz, _vp, _vp, _vp, _sz, _vp]),
    'crfconv_fps': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'crfconv_grid_subsample': (_i64, [_vp, _i64, _vp, _i, _vp, _i, _f, _vp, _vp, _vp, _i64]),
    'crfconv_grid_subsample_dev_workspace': (_sz, [_i64, _i, _i]),
    'crfconv_grid_subsample_dev': (_i64, [_vp, _i64, _vp, _i, _vp, _i, _f, _vp, _vp, _vp, _i64, _vp, _sz, _vp]),
    'crfconv_index_narrow': (_i, [_vp, _i64, _i64, _i, _i64, _vp, _vp, _vp, _vp]),
    'crfconv_index_narrow_sorted': (_i, [_vp, _i64, _i64, _i, _i64, _i, _vp, _vp, _vp, _vp]),
    'crfconv_reverse_csr_workspace': (_sz, [_i64, _i64]),
    'crfconv_reverse_csr': (_i, [_vp, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    'crfconv_reverse_csr_batched_workspace': (_sz, [_vp, _i]),
    'crfconv_reverse_csr_batched': (_i, [_vp, _i, _vp, _sz, _vp]),
    'crfconv_meanfield_forward': (_i, [_vp, _vp, _vp, _i, _i, _i64, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    'crfconv_meanfield_forward_u16': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i64, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    'crfconv_meanfield_forward_block_rows': (_i, [_i64, _i, _i, _i, _i]),
    'crfconv_meanfield_forward_block': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i64, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    'crfconv_meanfield_forward_block_stamps': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i64, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    'crfconv_block_locality': (_i, [_vp, _i64, _i, _i, _i, _vp, _vp]),
    'crfconv_meanfield_bwd_edge': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i64, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    'crfconv_meanfield_bwd_scatter': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i64, _i, _vp, _vp, _vp]),
    'crfconv_meanfield_backward_supported': (_i, [_i, _i, _i]),
    'crfconv_meanfield_backward_param_grads_inside': (_i, [_i]),
    'crfconv_meanfield_backward_workspace': (_sz, [_i64, _i, _i]),
    'crfconv_meanfield_backward': (_i, [_vp] * 7 + [_i, _i, _vp, _vp, _i, _i, _i64, _i, _vp, _vp, _i] + [_vp] * 11 + [_sz, _vp, _vp]),
    'crfconv_wide_similarity': (_i, [_vp, _vp, _i, _i, _i64, _i, _vp, _vp]),
    'crfconv_wide_aggregate': (_i, [_vp, _vp, _vp, _i, _i, _i64, _i, _vp, _vp]),
    'crfconv_wide_bwd_edge': (_i, [_vp, _vp, _vp, _i, _i, _i64, _i, _vp, _i, _vp]),
    'crfconv_wide_scatter': (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _vp, _i, _vp, _vp]),
    'crfconv_wide_similarity_bwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i64, _i, _vp, _vp, _vp]),
    'crfconv_similarity_bwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i64, _i, _vp, _vp, _vp]),
    'crfconv_similarity_bwd_scatter': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _i, _vp, _vp]),
    'crfconv_pointconv_workspace': (_sz, [_i64, _i, _i]),
    'crfconv_pointconv_moments_packed': (_i, [_vp, _vp, _vp, _i, _i64, ctypes.c_double, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'crfconv_pointconv_moments': (_i, [_vp, _vp, _vp, _i, _i64, _vp, _vp, _sz, _vp]),
    'crfconv_pointconv_stats': (_i, [_vp, _vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    'crfconv_pointconv_forward_uv': (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    'crfconv_pointconv_forward_uv_hosts': (_i, [_i, _i]),
    'crfconv_pointconv_forward_uv_hosting': (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp,
                                                  _vp, _vp, _i, _vp, _vp, _vp]),
    'crfconv_pointconv_combine': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _d, _vp, _vp, _f, _f, _i64, _i, _vp, _vp, _vp, _vp, _vp]),
    'crfconv_pointconv_bwd_reduce_uv': (_i, [_vp, _vp, _vp, _i64, _i, _vp, _vp, _vp, _d, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    'crfconv_pointconv_forward': (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    'crfconv_pointconv_bwd_reduce': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _sz, _vp]),
    'crfconv_pointconv_bwd_params': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp,
                                          _vp, _vp, _vp, _sz, _vp]),
    'crfconv_pointconv_bwd_dump': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'crfconv_pointconv_bwd_a1_workspace': (_sz, [_i64, _i]),
    'crfconv_pointconv_bwd_a1': (_i, [_vp, _vp, _vp, _i64, _i, _f, _vp, _vp, _sz, _vp]),
    'crfconv_pointconv_bwd_input': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    'crfconv_pointconv_bwd_input_reduce': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _i64, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _i,
                                                _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    'crfconv_pointconv_fold1': (_i, [_vp, _vp, _vp, _vp, ctypes.c_double, _vp, _vp, _f, _f, _i, _i, _vp, _vp, _vp, _vp]),
    'crfconv_pointconv_fold1_batched': (_i, [_vp, _i, _vp]),
    'crfconv_pointconv_fold1_bwd_batched': (_i, [_vp, _i, _vp]),
    'crfconv_pointconv_fold1_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'crfconv_pointconv_fold2': (_i, [_vp, _vp, _vp, _vp, ctypes.c_double, _vp, _vp, _f, _f, _i, _i, _vp, _vp, _vp, _vp]),
    'crfconv_pointconv_fold2_bwd': (_i, [_vp, _vp, _vp, _vp, ctypes.c_double, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'crfconv_linear_wgrad_workspace': (_sz, [_i64, _i, _i]),
    'crfconv_linear_wgrad': (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    'crfconv_random_subsets': (_i, [_vp, _vp, _vp, _vp, _i, _u64, _vp, _vp]),
    'crfconv_upindex_workspace': (_sz, [_i64, _i64]),
    'crfconv_upindex_from_table': (_i, [_vp, _vp, _vp, _vp, _i64, _i64, _i, _i64, _vp, _vp, _sz, _vp]),
    'crfconv_gather_rows_batched': (_i, [_vp, _vp, _vp, _i, _vp, _i, _i64, _i64, _i64, _vp]),
    'crfconv_argsort_codes_workspace': (_sz, [_i64, _i64]),
    'crfconv_argsort_codes': (_i, [_vp, _i64, _i64, _vp, _vp, _sz, _vp]),
    'crfconv_gridsync_workspace': (_sz, []),
    'crfconv_gridsync_fail_word': (_i, []),
    'crfconv_mlp_small_supported': (_i, [_i64, _i, _i]),
    'crfconv_mlp_small_workspace': (_sz, [_i64, _i]),
    'crfconv_mlp_small_forward': (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    'crfconv_mlp_small_forward_join': (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _f, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    'crfconv_ticket_bytes': (_sz, []),
    'crfconv_mlp_backward_supported': (_i, [_i64, _i, _i]),
    'crfconv_mlp_backward_workspace': (_sz, [_i64, _i, _i]),
    'crfconv_mlp_backward': (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    'crfconv_linear_forward_cat': (_i, [_vp, _vp, _i, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp]),
    'crfconv_mlp_dw_jobs': (_i, [_vp, _i, _vp]),
    'crfconv_mlp_dw_jobs_hosting': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    'crfconv_mlp_backward_add': (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    'crfconv_mlp_backward_cat': (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _f, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    'crfconv_bn_workspace': (_sz, [_i64, _i]),
    'crfconv_bn_forward': (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp, _f, _f, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    'crfconv_bn_backward': (_i, [_vp, _vp, _vp, _i64, _i, _i, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    'crfconv_bn_apply': (_i, [_vp, _i64, _i, _vp, _f, _vp, _vp]),
    'crfconv_pointconv_bwd_params_slabs': (_i, [_vp, _i64, _i, _vp, _vp, _vp]),
    'crfconv_pointconv_bwd_a1_nblk': (_i64, [_i64, _i]),
    'crfconv_reduce_jobs_f64': (_i, [_vp, _i, _vp]),
    'crfconv_pointconv_bwd_dump_jobs': (_i, [_vp, _i, _vp]),
    'crfconv_pointconv_bwd_a1_jobs': (_i, [_vp, _i, _vp]),
    'crfconv_pointconv_wide_params_supported': (_i, [_i64, _i, _i]),
    'crfconv_pointconv_wide_params_nblk': (_i64, [_i64, _i]),
    'crfconv_pointconv_wide_params_jobs': (_i, [_vp, _i, _vp]),
    'crfconv_gemm_jobs': (_i, [_vp, _i, _vp]),
    'crfconv_linear_wgrad_nblk': (_i, [_i64, _i, _i]),
    'crfconv_linear_wgrad_partial_jobs': (_i, [_vp, _i, _vp]),
    'crfconv_linear_wgrad_partial': (_i, [_vp, _vp, _i64, _i, _i, _i, _vp, _sz, _vp, _vp]),
    'crfconv_morton_codes': (_i, [_vp, _i64, _i64, _vp, _vp, _vp]),
    'crfconv_copy_jobs': (_i, [_vp, _i, _vp]),
    'crfconv_index_narrow_batched': (_i, [_vp, _i, _vp]),
    'crfconv_pointconv_moments_batched_workspace': (_sz, [_vp, _i]),
    'crfconv_pointconv_moments_batched': (_i, [_vp, _i, _vp, _sz, _vp]),
    'crfconv_reduce_jobs_both': (_i, [_vp, _i, _vp, _i, _vp]),
    'crfconv_reduce_jobs': (_i, [_vp, _i, _vp]),
    'crfconv_linear_forward_supported': (_i, [_i, _i]),
    'crfconv_linear_forward_stat_records': (_sz, [_i64]),
    'crfconv_linear_forward': (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _vp, _vp, _vp]),
    'crfconv_cat2': (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp]),
    'crfconv_split2': (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp]),
    'crfconv_add_i64': (_i, [_vp, _i64, _i64, _vp]),
    'crfconv_gate_mark': (_i, [_vp, _vp]),
    'crfconv_gate_wait': (_i, [_vp, _i, _vp]),
    'crfconv_mlp_small_backward_supported': (_i, [_i64, _i, _i]),
    'crfconv_gemm_stats_jobs': (_i, [_vp, _i, _vp]),
    'crfconv_bn_apply_from_records_jobs': (_i, [_vp, _i, _vp]),
    'crfconv_mlp_small_backward_jobs': (_i, [_vp, _i, _vp, _vp]),
    'crfconv_mlp_small_backward_jobs_one_launch': (_i, [_vp, _i, _vp, _vp, _vp]),
    'crfconv_mlp_small_backward_workspace': (_sz, [_i64, _i]),
    'crfconv_mlp_small_backward': (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    'crfconv_gemm_stat_records': (_sz, [_i64]),
    'crfconv_gemm_stats': (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp, _vp]),
    'crfconv_gemm_supported': (_i, [_i64, _i, _i]),
    'crfconv_gemm': (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _vp]),
    'crfconv_bn_apply_from_records': (_i, [_vp, _i64, _vp, _i64, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _f, _vp, _vp, _vp]),
    'crfconv_bn_coef_from_nrecords': (_i, [_vp, _i64, _i64, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp]),
    'crfconv_bn_coef_from_records': (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp]),
    'crfconv_softmax_ce_workspace': (_sz, [_i64]),
    'crfconv_softmax_ce_forward': (_i, [_vp, _vp, _vp, _i64, _i, _i64, _i64, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    'crfconv_softmax_ce_backward': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i64, _i64, _vp, _vp]),
    'crfconv_crf_matrices': (_i, [_vp, _i, _vp, _vp, _vp]),
    'crfconv_crf_matrices_backward': (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp]),
    'crfconv_crf_matrices_batched': (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    'crfconv_crf_matrices_backward_batched': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    'crfconv_add_lrelu': (_i, [_vp, _vp, _i64, _f, _vp, _vp]),
    'crfconv_bn_apply_add': (_i, [_vp, _i64, _i, _vp, _vp, _f, _vp, _vp]),
    'crfconv_bn_apply_dropout': (_i, [_vp, _i64, _i, _vp, _f, _f, _u64, _vp, _vp, _vp, _vp]),
    'crfconv_head_stats': (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp]),
    'crfconv_head_stat_records': (_sz, [_i64]),
    'crfconv_head_supported': (_i, [_i64, _i, _i, _i]),
    'crfconv_head_mask_words': (_sz, [_i64]),
    'crfconv_head_backward_workspace': (_sz, [_i64, _i, _i, _i]),
    'crfconv_head_forward': (_i, [_vp, _vp, _vp, _f, _f, _u64, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'crfconv_head_backward': (_i, [_vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _i64, _i, _i, _i] + [_vp] * 7 + [_sz, _vp]),
    'crfconv_linear_forward_dropout': (_i, [_vp, _vp, _i64, _i, _i, _i, _f, _u64, _vp, _vp, _vp]),
    'crfconv_dropout_backward': (_i, [_vp, _i64, _f, _u64, _vp, _vp, _vp]),
    'crfconv_add_lrelu_backward': (_i, [_vp, _vp, _i64, _f, _vp, _vp]),
    'crfconv_sgd_step': (_i, [_vp, _vp, _vp, _i64, _f, _f, _f, _f, _i, _i, _vp]),
    'crfconv_sgd_step_hyper': (_i, [_vp, _vp, _vp, _i64, _vp, _i, _i, _vp]),
    'crfconv_sgd_step_guarded': (_i, [_vp, _vp, _vp, _i64, _vp, _i, _i, _vp, _vp]),
    'crfconv_sgd_step_guarded_all': (_i, [_vp, _vp, _vp, _i64, _vp, _i, _i, _vp, _i, _vp, _vp]),
    'crfconv_sgd_guard_publish': (_i, [_vp, _i, _vp, _vp]),
    'crfconv_spd_inverse': (_i, [_vp, _i, _vp, _vp]),
    'crfconv_spd_inverse_wide': (_i, [_vp, _i, _vp, _vp]),
    'crfconv_neighbor_maxpool_forward': (_i, [_vp, _vp, _i, _i64, _i, _vp, _vp, _vp]),
    'crfconv_neighbor_maxpool_affine_forward': (_i, [_vp, _vp, _vp, _i, _i64, _i, _vp, _vp, _vp]),
    'crfconv_neighbor_maxpool_backward': (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _vp, _vp]),
    'crfconv_gather_rows': (_i, [_vp, _vp, _i64, _i, _vp, _vp]),
    'crfconv_gather_rows_backward': (_i, [_vp, _vp, _vp, _i64, _i, _vp, _vp]),
    'crfconv_meanfield_step': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i64, _i, _vp, _vp, _vp, _vp]),
    'crfconv_kernel_weights_forward': (_i, [_vp, _vp, _i, _vp, _i, _i, _i64, _vp, _vp]),
    'crfconv_kernel_weights_partials': (_sz, [_i64]),
    'crfconv_kernel_weights_backward': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i64, _vp, _vp, _vp, _vp]),
    'crfconv_confusion_accumulate': (_i, [_vp, _vp, _vp, _i64, _i, _i64, _i64, _vp, _vp, _vp]),
    'crfconv_vote_accumulate': (_i, [_vp, _vp, _vp, _i64, _i, _d, _vp, _i64, _vp, _vp]),
    'crfconv_vote_accumulate_counted': (_i, [_vp, _vp, _vp, _i64, _i, _d, _vp, _i64, _vp, _vp, _vp]),
    'crfconv_vote_fold': (_i, [_vp, _vp, _vp, _vp, _i64, _i, _d, _vp]),
    'crfconv_vote_project': (_i, [_vp, _vp, _i64, _i, _i64, _i, _vp, _vp, _vp]),
    'crfconv_argmin_workspace': (_sz, []),
    'crfconv_argmin_f64': (_i, [_vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    'crfconv_possibility_crop_workspace': (_sz, [_i64, _i64]),
    'crfconv_possibility_crop': (_i, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
}



class ReduceJob(ctypes.Structure):
    """crf_reduce_job of include/crfconv_amd.h."""
    _fields_ = [('partial', ctypes.c_void_p), ('out', ctypes.c_void_p), ('nblk', ctypes.c_int32), ('nslots', ctypes.c_int32)]


class CopyJob(ctypes.Structure):
    """crf_copy_job of include/crfconv_amd.h."""
    _fields_ = [('src', ctypes.c_void_p), ('dst', ctypes.c_void_p), ('nbytes', ctypes.c_int64)]


class MlpDwJob(ctypes.Structure):
    """crf_mlp_dw_job of include/crfconv_amd.h."""
    _fields_ = [('workspace', ctypes.c_void_p), ('coef', ctypes.c_void_p), ('dW', ctypes.c_void_p), ('M', ctypes.c_int64),
                ('Ci', ctypes.c_int32), ('Co', ctypes.c_int32)]


class PcDumpJob(ctypes.Structure):
    """crf_pc_dump_job of include/crfconv_amd.h."""
    _fields_ = [('x', ctypes.c_void_p), ('gout', ctypes.c_void_p), ('pos_src', ctypes.c_void_p), ('pos_tgt', ctypes.c_void_p),
                ('idx32', ctypes.c_void_p), ('K', ctypes.c_int32), ('m_tgt', ctypes.c_int64), ('d', ctypes.c_int32), ('A1', ctypes.c_void_p),
                ('b1', ctypes.c_void_p), ('W2', ctypes.c_void_p), ('slope', ctypes.c_float), ('ca', ctypes.c_void_p), ('cb', ctypes.c_void_p),
                ('cc', ctypes.c_void_p), ('h1', ctypes.c_void_p), ('gh2', ctypes.c_void_p), ('rel', ctypes.c_void_p)]


class PcWideJob(ctypes.Structure):
    """crf_pc_wide_job of include/crfconv_amd.h."""
    _fields_ = [('x', ctypes.c_void_p), ('gout', ctypes.c_void_p), ('pos_src', ctypes.c_void_p), ('pos_tgt', ctypes.c_void_p),
                ('idx32', ctypes.c_void_p), ('K', ctypes.c_int32), ('m_tgt', ctypes.c_int64), ('d', ctypes.c_int32), ('A1', ctypes.c_void_p),
                ('b1', ctypes.c_void_p), ('W2', ctypes.c_void_p), ('slope', ctypes.c_float), ('ca', ctypes.c_void_p), ('cb', ctypes.c_void_p),
                ('cc', ctypes.c_void_p), ('dw2_partial', ctypes.c_void_p), ('a1_partial', ctypes.c_void_p)]


class PcA1Job(ctypes.Structure):
    """crf_pc_a1_job of include/crfconv_amd.h."""
    _fields_ = [('gw', ctypes.c_void_p), ('h1', ctypes.c_void_p), ('rel', ctypes.c_void_p), ('n_edges', ctypes.c_int64), ('d', ctypes.c_int32),
                ('slope', ctypes.c_float), ('workspace', ctypes.c_void_p), ('workspace_bytes', ctypes.c_size_t)]


class GemmJob(ctypes.Structure):
    """crf_gemm_job of include/crfconv_amd.h."""
    _fields_ = [('A', ctypes.c_void_p), ('B', ctypes.c_void_p), ('C', ctypes.c_void_p), ('M', ctypes.c_int64), ('N', ctypes.c_int32),
                ('K', ctypes.c_int32)]


class GemmStatsJob(ctypes.Structure):
    """crf_gemm_stats_job of include/crfconv_amd.h."""
    _fields_ = [('A', ctypes.c_void_p), ('B', ctypes.c_void_p), ('M', ctypes.c_int64), ('N', ctypes.c_int32), ('K', ctypes.c_int32),
                ('C', ctypes.c_void_p), ('stat_rec', ctypes.c_void_p)]


class BnApplyJob(ctypes.Structure):
    """crf_bn_apply_job of include/crfconv_amd.h."""
    _fields_ = [('stat_rec', ctypes.c_void_p), ('nrec', ctypes.c_int64), ('x', ctypes.c_void_p), ('M', ctypes.c_int64), ('C', ctypes.c_int32),
                ('gamma', ctypes.c_void_p), ('beta', ctypes.c_void_p), ('run_mean', ctypes.c_void_p), ('run_var', ctypes.c_void_p),
                ('momentum', ctypes.c_float), ('eps', ctypes.c_float), ('skip', ctypes.c_void_p), ('slope', ctypes.c_float),
                ('coef', ctypes.c_void_p), ('y', ctypes.c_void_p)]


class MlpBwdJob(ctypes.Structure):
    """crf_mlp_bwd_job of include/crfconv_amd.h."""
    _fields_ = [('gA', ctypes.c_void_p), ('Y', ctypes.c_void_p), ('coef', ctypes.c_void_p), ('W', ctypes.c_void_p), ('addend', ctypes.c_void_p),
                ('M', ctypes.c_int64), ('Ci', ctypes.c_int32), ('Co', ctypes.c_int32), ('training', ctypes.c_int32), ('slope', ctypes.c_float),
                ('gY', ctypes.c_void_p), ('dX', ctypes.c_void_p), ('dgamma', ctypes.c_void_p), ('dbeta', ctypes.c_void_p),
                ('workspace', ctypes.c_void_p), ('workspace_bytes', ctypes.c_size_t)]


class Reduce64Job(ctypes.Structure):
    """crf_reduce64_job of include/crfconv_amd.h."""
    _fields_ = [('partial', ctypes.c_void_p), ('is_float', ctypes.c_int32), ('nblk', ctypes.c_int64), ('nslots', ctypes.c_int32),
                ('out', ctypes.c_void_p)]


class WgradJob(ctypes.Structure):
    """crf_wgrad_job of include/crfconv_amd.h."""
    _fields_ = [('G', ctypes.c_void_p), ('X', ctypes.c_void_p), ('M', ctypes.c_int64), ('Co', ctypes.c_int32), ('Ci', ctypes.c_int32),
                ('want_bias', ctypes.c_int32), ('workspace', ctypes.c_void_p), ('workspace_bytes', ctypes.c_size_t)]


class RevJob(ctypes.Structure):
    """crf_rev_job of include/crfconv_amd.h."""
    _fields_ = [('idx32', ctypes.c_void_p), ('E', ctypes.c_int64), ('m_src', ctypes.c_int64), ('rev_ptr', ctypes.c_void_p),
                ('rev_eid', ctypes.c_void_p)]


class NarrowJob(ctypes.Structure):
    """crf_narrow_job of include/crfconv_amd.h."""
    _fields_ = [('idx64', ctypes.c_void_p), ('B', ctypes.c_int64), ('n_tgt', ctypes.c_int64), ('K', ctypes.c_int32),
                ('n_src', ctypes.c_int64), ('sort_from', ctypes.c_int32), ('idx32', ctypes.c_void_p), ('idx16', ctypes.c_void_p),
                ('bad_count', ctypes.c_void_p)]


class MomentsJob(ctypes.Structure):
    """crf_moments_job of include/crfconv_amd.h."""
    _fields_ = [('pos_src', ctypes.c_void_p), ('pos_tgt', ctypes.c_void_p), ('idx32', ctypes.c_void_p), ('K', ctypes.c_int32),
                ('m_tgt', ctypes.c_int64), ('n_edges', ctypes.c_double), ('mean', ctypes.c_void_p), ('cov', ctypes.c_void_p),
                ('packed', ctypes.c_void_p), ('mean32', ctypes.c_void_p)]


class Fold1Job(ctypes.Structure):
    """crf_fold1_job of include/crfconv_amd.h."""
    _fields_ = [('W1', ctypes.c_void_p), ('gamma1', ctypes.c_void_p), ('beta1', ctypes.c_void_p), ('mom', ctypes.c_void_p),
                ('n_edges', ctypes.c_double), ('run_mean', ctypes.c_void_p), ('run_var', ctypes.c_void_p),
                ('momentum', ctypes.c_float), ('eps', ctypes.c_float), ('use_batch', ctypes.c_int32), ('d', ctypes.c_int32),
                ('A1', ctypes.c_void_p), ('b1', ctypes.c_void_p), ('aux1', ctypes.c_void_p)]


class Fold1BwdJob(ctypes.Structure):
    """crf_fold1_bwd_job of include/crfconv_amd.h."""
    _fields_ = [('W1', ctypes.c_void_p), ('gamma1', ctypes.c_void_p), ('mom', ctypes.c_void_p), ('aux1', ctypes.c_void_p),
                ('dA1b1', ctypes.c_void_p), ('eps', ctypes.c_float), ('use_batch', ctypes.c_int32), ('d', ctypes.c_int32),
                ('pad_', ctypes.c_int32), ('dW1', ctypes.c_void_p), ('dgamma1', ctypes.c_void_p), ('dbeta1', ctypes.c_void_p),
                ('dW2_f64', ctypes.c_void_p), ('dW2_f32', ctypes.c_void_p)]


_lib = None


class CrfConvError(RuntimeError):
    pass


def load():
    """Loads the shared library (once). Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CrfConvError(
                '%s not found: build it with `make -C crfconv_amd/csrc` (or __graft_entry__.build()). '
                'crfconv_amd has no CPU / PyTorch fallback.' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError here == header and library out of sync
            fn.restype = res
            fn.argtypes = args
        if lib.crfconv_abi_version() != 1:
            raise CrfConvError('libcrfconv_amd.so ABI version mismatch')
        _lib = lib
    return _lib


def last_error():
    return load().crfconv_last_error().decode('utf-8', 'replace')


def check(rc, what=''):
    """Raise on a negative status; returns rc otherwise (sizes / counts pass through)."""
    if rc < 0:
        raise CrfConvError('%s failed (%d): %s' % (what or 'libcrfconv_amd call', rc, last_error()))
    return rc


_FN = {}          # name -> bound foreign function (the eager path makes ~300 calls per training step: no lookup chain per call)


def call(name, *args):
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(load(), name)
    rc = fn(*args)
    if rc < 0:
        check(rc, name)
    return rc
