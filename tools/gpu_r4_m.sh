#!/bin/bash
# round 4, call m: which kernel-variant test leaves the process in a state in which the derived protein2genome set fails?
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4m; mkdir -p $OUT
P="tests/test_gpu_parity.py::test_find_score_and_path_match_reference_vectors[derived_protein2genome_end]"
for t in test_unpacked_region_kernels_match_reference_vectors test_general_kernels_match_reference_vectors test_blocking_kernels_of_both_variants \
         test_seeded_pairs_on_unpacked_and_general_kernels test_extreme_parameters_match_oracle test_c5_protein2genome_against_a_10mb_contig \
         test_find_path_over_regions_of_resident_pairs test_windowed_region_pass_matches_oracle test_window_kernel_on_two_and_four_waves_agree \
         test_windowed_and_one_pass_region_agree_at_full_size test_window_hop_budget_covers_paths_across_the_whole_window \
         test_packed_16_bit_score_pass_agrees_with_the_32_bit_pass test_staged_packed_score_pass_agrees_with_the_plain_one \
         test_packed_16_bit_region_windows_agree_with_the_32_bit_windows test_device_route_of_the_sub_alignments_gives_the_host_route_results \
         test_packed_16_bit_checkpoint_pass_agrees_with_the_32_bit_pass test_two_launch_lanes_give_the_one_lane_results test_memory_rule_on_the_device_is_the_host_rule; do
  timeout 600 python -m pytest "tests/test_gpu_kernel_variants.py" "$P" -m gpu -q -k "$t or derived_protein2genome_end" > $OUT/$t.log 2>&1
  echo "$t: $(tail -1 $OUT/$t.log)"
done
# memory left behind by an earlier allocation: a tensor of 0x5a bytes freed just before
python - > $OUT/pollute.log 2>&1 <<'P'
import torch, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
x = torch.full((8 << 30,), 0x5a, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); del x; torch.cuda.empty_cache()
import pytest
sys.exit(pytest.main(["tests/test_gpu_parity.py", "-m", "gpu", "-q", "-k", "derived_protein2genome"]))
P
echo "after polluted memory: $(tail -1 $OUT/pollute.log)"
