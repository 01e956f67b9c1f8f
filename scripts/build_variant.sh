#!/bin/bash
# tuning aid: a variant of the library with extra compile flags for the kernel translation units
#   scripts/build_variant.sh <name> <flags...>   ->  krakenuniq_amd/variants/libku_<name>.so   (select it with KU_LIB=<path>)
#   and krakenuniq_amd/variants/<name>/libkrakenuniq_amd.so, the same file under the product's name (for the executables:
#   LD_LIBRARY_PATH=krakenuniq_amd/variants/<name>, scripts/cli_probe.py LIB=)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../krakenuniq_amd/csrc"
mkdir -p ../variants /tmp/kuvar_$NAME
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include $*"
/opt/rocm/bin/hipcc $F -c ku_short.hip -o /tmp/kuvar_$NAME/ku_short.o &
/opt/rocm/bin/hipcc $F -c ku_kernels.hip -o /tmp/kuvar_$NAME/ku_kernels.o &
/opt/rocm/bin/hipcc $F -c ku_route.hip -o /tmp/kuvar_$NAME/ku_route.o &
/opt/rocm/bin/hipcc $F -c ku_report.hip -o /tmp/kuvar_$NAME/ku_report.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/libku_$NAME.so /tmp/kuvar_$NAME/ku_kernels.o /tmp/kuvar_$NAME/ku_short.o /tmp/kuvar_$NAME/ku_route.o \
  ku_sparse.o /tmp/kuvar_$NAME/ku_report.o ku_dbsort.o ku_setlcas.o ku_api.o ku_api_sparse.o ku_api_classify.o ku_api_rle.o ku_api_ooc.o ku_api_report.o ku_mgpu.o ku_host.o -ldl -lpthread
mkdir -p ../variants/$NAME && cp ../variants/libku_$NAME.so ../variants/$NAME/libkrakenuniq_amd.so
echo built ../variants/libku_$NAME.so
