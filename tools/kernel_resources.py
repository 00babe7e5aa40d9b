#!/usr/bin/env python3
"""Prints VGPR / SGPR / scratch / spill counts per kernel from the code-object metadata of a hipcc -S output.
   The frame-loop kernels are compiled one FFT size per translation unit (spectroplot-js_amd/Makefile), e.g. for n = 1024:
     cd spectroplot-js_amd && hipcc -O3 -std=c++17 -ffp-contract=off -fno-fast-math -DSP_INST_FRAMES_LOG2N=10 --offload-arch=gfx950 \
        --offload-device-only -S -o /tmp/f10.s csrc/sp_inst_frames.hip && python3 ../tools/kernel_resources.py /tmp/f10.s k_frames
   (tools/spills.sh prints the spill table of every built variant straight from the objects.)"""
import re, sys
txt = open(sys.argv[1]).read()
for blk in re.findall(r"- \.agpr_count:.*?\.wavefront_size:\s*\d+", txt, flags=re.S):
    g = lambda k: re.search(r"\.%s:\s*(\S+)" % k, blk)
    name = g("name").group(1)
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    print("%-95s vgpr=%s sgpr=%s scratch=%s vspill=%s sspill=%s lds=%s" % (name[:95], g("vgpr_count").group(1), g("sgpr_count").group(1),
          g("private_segment_fixed_size").group(1), g("vgpr_spill_count").group(1), g("sgpr_spill_count").group(1), g("group_segment_fixed_size").group(1)))
