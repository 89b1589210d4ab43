#!/bin/bash
# Round 5 (GPU box): random geometries through the halo-kernel probes (tools/ig_halo_probe.cpp, tools/stem_halo_probe.cpp): every
# case compares the halo kernel with the conv_tile launch it replaces bit for bit.   bash tools/halo_soak.sh [cases] > log
# Prints one line per case and a summary; exit code 1 if any case differs.
N=${1:-60}; RANDOM=${2:-12345}; bad=0; ran=0
for i in $(seq 1 $N); do
  H=$((16 + RANDOM % 120)); F=$((1 + RANDOM % 5))
  case $((RANDOM % 5)) in
    0) a="$F $H 64 7 2 3 2";;                                        # ResNet-like image stem, 64 channels
    1) a="$F $H $((16 * (1 + RANDOM % 3))) 7 2 3 2";;                # 16 / 32 / 48 channels
    2) a="$F $H 64 3 2 0 2";;                                        # SqueezeNet-like 3x3/2
    3) a="$((1 + RANDOM % 2)) $H 32 7 2 3 2 5 2 2 $((3 + RANDOM % 14))";;   # I3D-like video stem (24 class rows, frame taps)
    4) a="$((1 + RANDOM % 2)) $H $((4 * (1 + RANDOM % 2))) 7 2 3 2 5 2 2 $((3 + RANDOM % 14))";;   # quad-row order, 4 / 8 channels
  esac
  out=$(timeout 120 tools/igh_probe $a 2>&1 | grep bitwise); ran=$((ran + 1))
  echo "igh $a : $out"; case "$out" in "bitwise: 0 of"*) ;; *) bad=$((bad + 1));; esac
done
for i in $(seq 1 $((N / 2))); do
  a="$((1 + RANDOM % 3)) $((5 + RANDOM % 30)) $((16 + 2 * (RANDOM % 60))) $((16 + 4 * (RANDOM % 30))) $((4 * (1 + RANDOM % 2))) 2"
  out=$(timeout 120 tools/sth_probe $a 2>&1 | grep bitwise); ran=$((ran + 1))
  echo "sth $a : $out"; case "$out" in "bitwise: 0 of"*) ;; *) bad=$((bad + 1));; esac
done
for i in $(seq 1 $((N / 2))); do      # the wide image stem (values and gate words): output width a multiple of 16
  a="$((1 + RANDOM % 4)) 1 $((16 + 2 * (RANDOM % 100))) $((32 * (1 + RANDOM % 7))) 64 2"
  out=$(timeout 120 tools/sth_probe $a 2>&1 | grep bitwise); ran=$((ran + 1))
  echo "sth $a : $out"; case "$out" in "bitwise: 0 of"*", 0 of"*) ;; *) bad=$((bad + 1));; esac
done
echo "halo soak: $ran cases, $bad not bit-identical"
[ $bad -eq 0 ]
