# same-box A/B of variant libraries: tools/ab.sh "<workloads>" lib1 lib2 ...   (two rounds)
set -e
W="$1"; shift
for round in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    DSPX_LIB=$PWD/dsp_amd/lib/$lib timeout -k 10 150 python tools/qb.py $W
  done
done
