for e in "AGP_CHOL_GROUP=1" "AGP_CHOL_GROUP=4 AGP_CHOL_LOOKAHEAD=0" "AGP_CHOL_GROUP=4" "AGP_CHOL_GROUP=8"; do
echo "== $e"; env $e timeout 300 python tools/bench_potrf.py < /dev/null 2>&1 | grep -v "loop not unrolled" | tail -12
done
