for ct in 0.25 0.5 1.0 2.0 4.0; do echo "== ct $ct"; GBL_SAH_CT=$ct timeout -k 10 200 python tools/variant_bench.py libgoblin_hip.so 2>&1 | grep scene; done
