# What is raster_fwd short of?  N extra operations of one kind per splat (results unused, same dynamic work), occupancy caps, and the
# rare-path branch removed — all replayed on the same arguments inside one process (tools/ab_kernels.py).
cd "${GRAFT_REPO_ROOT:-.}"
L=""; for v in b0 pk pk_v4 pk_v8 pk_e2 pk_s8 pk_s16 pk_b2 pk_l2 pk_nobr pk_w6 pk_w5 pk_w4; do L="$L,gpurun_ab/lib_$v.so"; done
python tools/ab_kernels.py --entry dnsplat_raster_fwd --libs ${L#,} --workload c2 --rounds 12 2>&1 | grep -v amdgpu
