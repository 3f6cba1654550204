# Same-box A/B of two builds of libpunet_hip.so (PUNET_HIP_LIB selects the library): build the baseline into
# probabilistic_domain_adaptation_amd/libpunet_hip_prev.so, the candidate into libpunet_hip.so, then run this on the GPU box.
# Boxes of the pool differ by up to ~4 %, so only numbers from ONE call are comparable.
for r in 1 2; do
for l in libpunet_hip_prev.so libpunet_hip.so; do echo "== $l"; PUNET_HIP_LIB=$PWD/probabilistic_domain_adaptation_amd/$l timeout -k 10 200 python tools/bench_x6_layers.py --rounds 3 | tail -1 || exit 1; done
done
