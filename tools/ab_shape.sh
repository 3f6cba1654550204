# Same-box A/B of the two MFMA shapes of the 3x3 forward / input-gradient kernels (one library, $PUNET_X6_MFMA_SHAPE):
# 32 = conv3x3_x6_kernel (v_mfma_f32_32x32x16_bf16, register-staged weights), 16 = conv3x3_x6b_kernel (16x16x32, LDS-DMA).
for r in 1 2; do
echo "== 32x32x16"; PUNET_X6_MFMA_SHAPE=32 timeout -k 10 200 python tools/bench_x6_layers.py --rounds 3 | tail -1 || exit 1
echo "== 16x16x32"; PUNET_X6_MFMA_SHAPE=16 timeout -k 10 200 python tools/bench_x6_layers.py --rounds 3 | tail -1 || exit 1
done
