cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" timeout 200 python tools/lde_throughput.py 20x78 19x148 19x314 18x114 > gpurun_out/x_$tag.log 2>&1; echo "== $tag"; grep "2^" gpurun_out/x_$tag.log; }
run base LURKHIP_X=0
run stag1 LURKHIP_LDE_STAGGER=1
run stag2 LURKHIP_LDE_STAGGER=2
run c4 LURKHIP_LDE_IO_LOG_C=4
run c4stag1 LURKHIP_LDE_IO_LOG_C=4 LURKHIP_LDE_STAGGER=1
run c4stag2 LURKHIP_LDE_IO_LOG_C=4 LURKHIP_LDE_STAGGER=2
run c4stag3 LURKHIP_LDE_IO_LOG_C=4 LURKHIP_LDE_STAGGER=3
run batch1 LURKHIP_LDE_SLAB_BATCH=1
run batch2 LURKHIP_LDE_SLAB_BATCH=2
run batch4 LURKHIP_LDE_SLAB_BATCH=4
