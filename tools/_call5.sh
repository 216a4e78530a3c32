cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/dma_conv_pattern.hip -o /tmp/dma_conv && /tmp/dma_conv
