for r in 1 2 3; do
REPS=40 python tools/dense_mixture_timing.py 65536 2>&1 | grep -E "^gmm50_pis_headline  |cfg3"
REPS=40 SDEH_LIBRARY=$PWD/prof_tmp/libsdeh_sgprall.so python tools/dense_mixture_timing.py 65536 2>&1 | grep -E "^gmm50_pis_headline  |cfg3" | sed 's/^/SGPR  /'
done
