#!/bin/bash
# round 5, call 23: MCRT_WF_DEAL_XCD=1 - the trace kernel's workgroups of one XCD take neighbouring blocks of the ray queue (one L2 sees
# 2048+ consecutive rays instead of every eighth block of 64) - C3 / C4 / spaceship probes, with 64- and 256-ray blocks
mkdir -p gpurun_out/r05
L=gpurun_out/r05/ab_trace_deal_xcd.log
: > $L
timeout 600 python tools/ab_probe.py c3 --sqrtspp 8 --steps 2 "base:" "xcd:MCRT_WF_DEAL_XCD=1" "xcd_deal8:MCRT_WF_DEAL_XCD=1,MCRT_WF_DEAL=8" "base:" "xcd:MCRT_WF_DEAL_XCD=1" 2>&1 | grep '^{' | cut -c1-150 | sed "s/^/c3 /" | tee -a $L
timeout 600 python tools/ab_probe.py c4 --sqrtspp 4 --steps 2 "base:" "xcd:MCRT_WF_DEAL_XCD=1" "xcd_deal8:MCRT_WF_DEAL_XCD=1,MCRT_WF_DEAL=8" "base:" "xcd:MCRT_WF_DEAL_XCD=1" 2>&1 | grep '^{' | cut -c1-150 | sed "s/^/c4 /" | tee -a $L
timeout 600 python tools/ab_probe.py spaceship --steps 3 "base:" "xcd:MCRT_WF_DEAL_XCD=1" "base:" "xcd:MCRT_WF_DEAL_XCD=1" 2>&1 | grep '^{' | cut -c1-150 | sed "s/^/spaceship /" | tee -a $L
