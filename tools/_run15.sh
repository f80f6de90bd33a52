hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/store_patterns.hip -o /tmp/store_patterns 2>/dev/null && /tmp/store_patterns
