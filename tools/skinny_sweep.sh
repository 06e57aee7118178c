for nc in 256 128 64 32; do echo "NC=$nc"; LDETR_SKINNY_NC=$nc timeout 300 python tools/bench_engine.py 16 2>&1 | grep "1x1" | cut -c1-115; done
