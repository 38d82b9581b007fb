# Round-5 parity soak on the GPU box (new since round 2: gen_rel.hip, the homography scorer on the matrix cores, the
# wider fp16 slack of the P3P filter, groups of 128): every mode of tests/parity_soak.py (single-problem entry points) and
# tests/parity_soak_batch.py (grouped entry point pl_ransac_batch).  Summary committed as profiles/r05_parity_soak.md.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_soak
mkdir -p $O
( echo "## plain (12..3000 correspondences)"; timeout 600 python tests/parity_soak.py 150 2601
  echo "## 7..40 correspondences"; SOAK_NMIN=7 SOAK_NMAX=40 timeout 600 python tests/parity_soak.py 150 2602
  echo "## SOAK_FUZZ (odd option values)"; SOAK_FUZZ=1 SOAK_NMAX=800 timeout 600 python tests/parity_soak.py 80 2603
  echo "## SOAK_FUZZ2 (bundle options, camera models)"; SOAK_FUZZ2=1 SOAK_NMAX=800 timeout 600 python tests/parity_soak.py 80 2604
  echo "## SOAK_FUZZ3 (degraded data)"; SOAK_FUZZ3=1 SOAK_NMAX=800 timeout 600 python tests/parity_soak.py 80 2605
  echo "## SOAK_FUZZ4 (warm starts, real_focal_check)"; SOAK_FUZZ4=1 SOAK_NMAX=800 timeout 600 python tests/parity_soak.py 80 2606
  echo "## SOAK_RANSAC (ransac_* entry points)"; SOAK_RANSAC=1 timeout 600 python tests/parity_soak.py 100 2607
  echo "## 2000..12000 correspondences"; SOAK_NMIN=2000 SOAK_NMAX=12000 timeout 900 python tests/parity_soak.py 40 2608
) > $O/soak.txt 2>&1
( echo "## pl_ransac_batch: groups of 16, 4 in flight, 12..6000 correspondences"; timeout 900 python tests/parity_soak_batch.py 120 2901 16 4
  echo "## pl_ransac_batch: groups of 128, 2 in flight, 256 MB arena (long runs cut into several batches)"; POSELIB_AMD_GROUP_ARENA_MB=256 timeout 900 python tests/parity_soak_batch.py 100 2902 128 2
  echo "## pl_ransac_batch: groups of 5, 8 in flight, 7..60 correspondences"; SOAK_NMIN=7 SOAK_NMAX=60 timeout 900 python tests/parity_soak_batch.py 120 2903 5 8 ) > $O/soak_batch.txt 2>&1
cat $O/soak.txt $O/soak_batch.txt | grep -v "^$" | cut -c1-220
