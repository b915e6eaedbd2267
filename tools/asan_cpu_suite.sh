#!/bin/bash
# Runs the CPU test suite (-m "not gpu") with the HOST side of libsr_yolo2.so (cfg parser, weights I/O, engine
# planning, detection / evaluation helpers, the C++ Detector) and the oracle compiled with AddressSanitizer +
# UndefinedBehaviorSanitizer.  Device code is untouched (GPU sanitizers are not available on the pool); the prebuilt
# kernel objects of the normal build are linked in.  Usage: tools/asan_cpu_suite.sh   (from the repo root, after `make`)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$(mktemp -d)
C=$ROOT/sr_object_detection_amd/csrc
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -O1 -g -fPIC"
g++ $SAN -std=c++17 -c $C/y2_imgfile.cpp -o $OUT/y2_imgfile.o
for f in y2_cfg y2_weights y2_engine y2_detect y2_eval y2_comm y2_feed; do
    gcc $SAN -ffp-contract=off -std=gnu11 -I$ROOT/include -I$C/host -c $C/host/$f.c -o $OUT/$f.o
done
g++ $SAN -std=c++17 -I$ROOT/include -I$C/host -c $C/yolo_v2_class.cpp -o $OUT/yolo_v2_class.o
g++ -shared -fPIC -fsanitize=address,undefined -o $OUT/libsr_yolo2.so $C/build/y2_runtime.o $C/build/y2_conv.o $C/build/y2_conv_f16.o \
    $C/build/y2_layers.o $C/build/y2_layers_f16.o $C/build/y2_detect_dev.o $C/build/y2_image.o $OUT/*.o -L/opt/rocm/lib -lamdhip64 -lm -lstdc++ -ldl
cp $ROOT/oracle/liby2oracle.so $OUT/liby2oracle.orig
trap 'cp $OUT/liby2oracle.orig $ROOT/oracle/liby2oracle.so; touch $ROOT/oracle/liby2oracle.so' EXIT
gcc $SAN -fopenmp -ffp-contract=off -shared -o $ROOT/oracle/liby2oracle.so $ROOT/oracle/y2_oracle.c -lm
cd $ROOT
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 \
    UBSAN_OPTIONS=print_stacktrace=1 Y2_LIB=$OUT/libsr_yolo2.so python -m pytest tests -q -m "not gpu" -p no:cacheprovider
