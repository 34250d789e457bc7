#!/bin/bash
# usage: scratch/seq_view.sh <trace_sequence.txt> [from] [to]  -- index, start, duration, workgroups, kernel of the launches of one replayed step
awk -v a=${2:-0} -v b=${3:-9999} '$1>=a && $1<=b {n=""; for(i=14;i<=NF;i++) n=n" "$i; printf "%3d %8.1f %6.1f wg=%-6s%s\n", $1, $3, $5, $8, n}' $1 | cut -c1-120
