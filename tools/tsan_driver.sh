#!/bin/bash
# ThreadSanitizer over the host driver's threaded paths (folds / ensemble members at once) on the CPU oracle backend:
# host code only, no GPU.  bash tools/tsan_driver.sh   (prints TSAN reports, if any, and the run's return codes)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
B=$(mktemp -d)
F="-O1 -g -fsanitize=thread -fPIC"
gcc $F -std=c11 -ffp-contract=off -fopenmp -D_POSIX_C_SOURCE=199309L -c $R/oracle/klr_oracle.c -o $B/o1.o
gcc $F -std=c11 -D_POSIX_C_SOURCE=199309L -c $R/oracle/klr_synth.c -o $B/o2.o
gcc $F -std=c11 -D_POSIX_C_SOURCE=199309L -I$R/oracle -c $R/oracle/klr_oracle_backend.c -o $B/o3.o
g++ $F -std=c++17 -c $R/kmerlr_amd/csrc/klr_driver.cpp -o $B/o4.o
g++ $F -std=c++17 -I$R/tools $R/tools/tsan_driver_harness.cpp $B/o1.o $B/o2.o $B/o3.o $B/o4.o -o $B/harness -lm -lgomp -lpthread
TSAN_OPTIONS="halt_on_error=0" $B/harness
rm -rf $B
