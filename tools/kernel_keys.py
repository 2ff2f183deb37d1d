"""Keys under which a traced kernel launch is aggregated (tools/profile_round.sh, tools/summarise_profile.py, bench.py).

A launch counts towards (1) its kernel FAMILY (`mlp_gemm_kernel`: every instantiation — what the kernel-stats tables show),
(2) for template families whose instantiations are different entry points, its INSTANTIATION (`mlp_gemm_kernel<PRO_BNRELU,
EPI_STATS>`), and (3) the C-ABI ENTRY POINT it was launched by (`entry:pn2_mlp_gemm`), so that `roofline.traffic` (PMC bytes
per launch) and `alg_bytes_per_launch` (bench.py, per entry point) describe the same population of launches
(the family average mixed the pooled and first-layer variants into `pn2_mlp_gemm`)."""
import re

PRO = {0: "PRO_NONE", 1: "PRO_BNRELU", 2: "PRO_GY", 3: "PRO_POOLG", 4: "PRO_FIRST", 5: "PRO_LIFT"}      # csrc/mlp_common.h
EPI = {0: "EPI_NONE", 1: "EPI_STATS", 2: "EPI_MASK", 3: "EPI_POOL", 4: "EPI_MASKL"}

ENTRY_OF_FAMILY = {
    "mlp_wgrad_kernel": "pn2_mlp_wgrad", "mlp_bwd_fused_kernel": "pn2_mlp_bwd_fused", "mlp_bwd_fused2_kernel": "pn2_mlp_bwd_fused_fold",
    "mlp_bwd_first_kernel": "pn2_mlp_bwd_fused_fold_first", "pool_bwd64_kernel": "pn2_pool_bwd", "pool_bwd128_kernel": "pn2_pool_bwd",
    "bn_relu_rows_max_kernel": "pn2_bn_relu_rows_max", "group_concat_rows_wide4_kernel": "pn2_group_concat_rows",
    "group_concat_rows_narrow_kernel": "pn2_group_concat_rows", "group_concat_rows_kernel": "pn2_group_concat_rows",
    "group_rows_grad_csr_kernel": "pn2_group_rows_grad", "group_rows_grad_kernel": "pn2_group_rows_grad",
    "mlp_wgrad_bf16_kernel": "pn2_mlp_wgrad_bf16", "mlp_bwd_bf16x_kernel": "pn2_mlp_bwd_bf16",
    "bn_relu_rows_max_bf16_v8_kernel": "pn2_bn_relu_rows_max_bf16", "group_concat_rows_bf16_wide8_kernel": "pn2_group_concat_rows_bf16",
    "bq_fused_group_kernel": "pn2_ball_query_group", "bq_slab_query_kernel": "pn2_ball_query", "bq_slab_build_kernel": "pn2_ball_query",
    "fps_multi_kernel": "pn2_furthest_point_sampling", "fps_coop_kernel": "pn2_furthest_point_sampling",
    "fps_resident_kernel": "pn2_furthest_point_sampling", "fps_order_m_kernel": "pn2_furthest_point_sampling", "fps_order_check_kernel": "pn2_furthest_point_sampling", "fps_bucket_kernel": "pn2_furthest_point_sampling",
    "gcn_linear_kernel": "pn2_gcn_linear", "gcn_bn_bwd_kernel": "pn2_gcn_linear_grad_w", "gcn_wgrad_kernel": "pn2_gcn_linear_grad_w",
    "gcn_linear_grad_x_kernel": "pn2_gcn_linear_grad_x",
    "group_lift_stats_kernel": "pn2_group_lift_rows", "lift_points_kernel": "pn2_lift_points",
    "inv_cloud_kernel": "pn2_group_inverse_index", "inv_keys_kernel": "pn2_group_inverse_index", "inv_ptr_kernel": "pn2_group_inverse_index",
    "three_interpolate_rows_grad_csr_kernel": "pn2_three_interpolate_rows_grad", "three_interpolate_rows_grad_kernel": "pn2_three_interpolate_rows_grad",
    "pool_bwd_prep_kernel": "pn2_pool_bwd_prep", "bn_relu_bwd_prep_kernel": "pn2_bn_relu_bwd_prep",
    "x3_pack_kernel": "pn2_x3_pack_weight", "x3_pack_first_kernel": "pn2_x3_pack_first",
    "group_points_grad_csr_kernel": "pn2_group_points_grad", "interp_grad_csr_lds_kernel": "pn2_three_interpolate_grad",
}


def family(name: str) -> str:
    m = re.search(r"([A-Za-z_0-9]+_kernel)", name)
    return m.group(1) if m else "other"


def keys(name: str):
    fam = family(name)
    out = [fam]
    if fam == "mlp_gemm_kernel":
        m = re.search(r"mlp_gemm_kernel<\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+)", name)
        if m:
            pro, epi = int(m.group(4)), int(m.group(5))
            out.append(f"mlp_gemm_kernel<{PRO.get(pro, pro)},{EPI.get(epi, epi)}>")
            out.append("entry:" + ("pn2_mlp_gemm_pool" if epi == 3 else "pn2_mlp_gemm_first" if pro == 4 else "pn2_mlp_gemm"))
    elif fam in ("group_lift_rows_kernel", "group_lift_rows_grad_kernel", "group_lift_rows_grad_heavy_kernel"):
        # template <R, BF>: the bf16-row instantiations are launched by the _bf16 entry points
        bf = re.search(r"_kernel<\s*\d+,\s*true", name) is not None
        out.append("entry:" + ("pn2_group_lift_rows" if fam == "group_lift_rows_kernel" else "pn2_group_lift_rows_grad") + ("_bf16" if bf else ""))
    elif fam == "mlp_bwd_bf16_kernel":
        # template <NTN, KTK, GMODE, FOLD, RECOMP, FY>: the fold / re-forming instantiations are launched by their own entry points
        m = re.search(r"mlp_bwd_bf16_kernel<\s*\d+,\s*\d+,\s*\d+,\s*(true|false)(?:,\s*(true|false))?(?:,\s*(true|false))?", name)
        fold, recomp, fy = (m.group(1) == "true", m.group(2) == "true", m.group(3) == "true") if m else (False, False, False)
        out.append("entry:" + ("pn2_mlp_bwd_bf16_pool" if recomp else "pn2_mlp_bwd_bf16_fold_first" if fy else
                               "pn2_mlp_bwd_bf16_fold" if fold else "pn2_mlp_bwd_bf16"))
    elif fam == "mlp_gemm_bf16_kernel":
        m = re.search(r"mlp_gemm_bf16_kernel<\s*\d+,\s*\d+,\s*(\d+),\s*(\d+)", name)
        pro, epi = (int(m.group(1)), int(m.group(2))) if m else (0, 0)
        out.append("entry:" + ("pn2_mlp_gemm_pool_bf16" if epi == 3 else "pn2_mlp_gemm_first_bf16" if pro == 4 else "pn2_mlp_gemm_bf16"))
    elif fam == "sa_eval_kernel":
        # template <IN, KA, KB, WAVES, PRO, EPI> (csrc/x3_chain.hip): IN 0 / 1 = the eval-mode SA level, IN 2 = the f32x3 training GEMM
        m = re.search(r"sa_eval_kernel<\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+)(?:,\s*(\d+),\s*(\d+))?", name)
        mode = int(m.group(1)) if m else 0
        if m:
            out.append(f"sa_eval_kernel<IN{mode},K{16 * int(m.group(2))},mid{16 * int(m.group(3))},pro{m.group(5) or 0},epi{m.group(6) or 0}>")
        # (IN 0 with the store + sums epilogue is pn2_x3_gemm_first: bench.py lists it under the entry it replaces)
        first = m is not None and mode == 0 and (m.group(6) or "0") == "1"
        out.append("entry:" + ("pn2_x3_gemm" if mode == 2 else "pn2_mlp_gemm_first" if first else "pn2_sa_eval_x3"))
    elif fam == "prep_vec_kernel":
        # template <POOLED>: true = pn2_pool_bwd_prep (and its segment-table form), false = pn2_bn_relu_bwd_prep
        out.append("entry:" + ("pn2_pool_bwd_prep" if re.search(r"prep_vec_kernel<\s*true", name) else "pn2_bn_relu_bwd_prep"))
    elif fam in ENTRY_OF_FAMILY:
        out.append("entry:" + ENTRY_OF_FAMILY[fam])
    return out
