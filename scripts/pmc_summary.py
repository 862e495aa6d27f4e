#!/usr/bin/env python3
"""Sums a rocprofv3 --pmc counter over the dispatches of the kernels of interest.
usage: pmc_summary.py <counter_collection.csv> <COUNTER>  -> JSON {kernel-prefix: {dispatches, sum}}"""
import collections
import csv
import json
import sys

path, counter = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(path)):
    if r["Counter_Name"] != counter:
        continue
    name = r["Kernel_Name"]
    for key in ("pair_join_bf16_kernelILi8ELi0E", "pair_join_bf16_kernelILi8ELi1E", "pair_join_bf16_kernelILi8ELi2E", "pair_join_bf16_kernel<8, 0>", "pair_join_bf16_kernel<8, 1>", "pair_join_bf16_kernel<8, 2>", "pair_join_mfma32_kernel", "theta_bf16_kernel<8, false>", "theta_bf16_kernel<8, true>", "rerank_join_d128_kernel", "rerank_join_kernel", "pair_topk_mfma32_kernel<4, 2, 1, asx::GraphEpi32>", "pair_topk_mfma32_kernel<4, 2, 1, asx::SearchEpi32>",
                "lambda_kernel_v3", "rerank_kernel<asx::GraphDistPolicy", "write_rows_kernel"):
        if key in name:
            acc[key][0] += 1
            acc[key][1] += float(r["Counter_Value"])
print(json.dumps({k: {"dispatches": v[0], counter: v[1]} for k, v in acc.items()}, indent=1))
