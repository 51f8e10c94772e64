#!/usr/bin/env python3
"""Writes a synthetic batch for tools/class_rotate_probe.cpp: <prefix>.chars (position-major, B x M bytes) and <prefix>.lens (u32).
usage: dump_batch.py <prefix> <B> <M> <regex23|regex1>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import halo2_regex_amd as hra
from halo2_regex_amd import synth
pre, B, M, which = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
gen = synth.regex23_planted if which == "regex23" else synth.regex1_planted
chars, lens = gen(B, M - 1, seed=0, stride=M)
hra.chars_to_position_major(chars).tofile(pre + ".chars")
lens.astype(np.uint32).tofile(pre + ".lens")
