for r in 1 2; do
echo "== default 32x32x16"; PUNET_X6_MFMA_SHAPE=32 timeout -k 10 200 python tools/bench_x6_layers.py --rounds 3 | tail -1
echo "== 16x16x32 (DMA weights)"; PUNET_X6_MFMA_SHAPE=16 timeout -k 10 200 python tools/bench_x6_layers.py --rounds 3 | tail -1
done
