/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DWISE_BUILD_FLAGS="-O3 --offload-arch=gfx950 -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops" -c /root/repo/wise_amd/csrc/htsat_frontend.hip -o /root/repo/wise_amd/lib/obj/htsat_frontend.o -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops
# build.py 73eb202ec08d3603
