#!/bin/bash
# TEST INFRASTRUCTURE ONLY.  Builds the UNMODIFIED reference (szcompressor/sz under /root/reference, read where it lies) into
# oracle/_ref/ (git-ignored; the built .so travels to the GPU box with the tree, the sources never do):
#   libSZ.so      the library the parity pin is recorded from (tools/record_reference_outputs.py, tools/ref_diff_fuzz.py) and the
#                 timed CPU baseline of bench.py (`cpu_baseline.kind = "reference"`)
#   libSZ_omp.so  the same sources with -fopenmp: the reference's OpenMP variant (SZ_compress_float_3D_MDQ_openmp, sz_omp.c)
#   libSZ_omp_O1.so  the same at -O1: the reference's DOUBLE OpenMP entry point runs off the end of a non-void function, which gcc -O3 turns
#                 into a trap; at -O1 it returns (round 4: what the float64 pin of the OpenMP container is recorded from; -O1 does not
#                 contract or reassociate either, the arithmetic is the same)
# One plain compiler line per library: gcc -O3 for baseline x86-64 (no FMA contraction: what the reference's own CMake Release
# build gives), the vendored zstd 1.3.5 and zlib 1.2.11 compiled in.  config.h is the reference's own template
# config.h.cmake with its four HAVE_* switches set (sys/time.h, unistd.h, clock_gettime, gettimeofday: all present on Linux) --
# exactly what `cmake` writes on this platform; nothing of the reference is edited or copied into the repository.
set -e
REF=${REF:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
if [ ! -d "$REF/sz/src" ]; then echo "build_ref.sh: $REF is not here (the GPU box): nothing to build, using what is in $OUT" >&2; exit 0; fi
mkdir -p "$OUT"
sed -E 's/^#cmakedefine (HAVE_[A-Z_]+) @.*@/#define \1 1/' "$REF/config.h.cmake" > "$OUT/config.h"
SRC=$(ls "$REF"/sz/src/*.c)
ZSTD=$(ls "$REF"/zstd/common/*.c "$REF"/zstd/compress/*.c "$REF"/zstd/decompress/*.c)
ZLIB=$(ls "$REF"/zlib/*.c)
INC="-I$OUT -I$REF/sz/include -I$REF/zstd -I$REF/zstd/common -I$REF/zlib"
CC=${CC:-gcc}
FLAGS="-O3 -fPIC -shared -w -std=gnu99 -DNDEBUG"
stamp="$OUT/.built_from"
sig="$(cd "$REF" && (git rev-parse HEAD 2>/dev/null || find sz/src zstd zlib -name '*.[ch]' | sort | xargs cat | md5sum | cut -d' ' -f1))"
if [ -f "$OUT/libSZ.so" ] && [ -f "$OUT/libSZ_omp.so" ] && [ -f "$OUT/libSZ_omp_O1.so" ] && [ "$(cat "$stamp" 2>/dev/null)" = "$sig" ]; then exit 0; fi
$CC $FLAGS $INC -o "$OUT/libSZ.so" $SRC $ZSTD $ZLIB -lm
$CC $FLAGS -fopenmp $INC -o "$OUT/libSZ_omp.so" $SRC $ZSTD $ZLIB -lm
$CC ${FLAGS/-O3/-O1} -fopenmp $INC -o "$OUT/libSZ_omp_O1.so" $SRC $ZSTD $ZLIB -lm
echo "$sig" > "$stamp"
echo "built $OUT/libSZ.so, $OUT/libSZ_omp.so and $OUT/libSZ_omp_O1.so"
