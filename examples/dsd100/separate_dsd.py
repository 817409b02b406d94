#!/usr/bin/env python3
"""Drop-in for the reference's examples/dsd100/separate_dsd.py, running on MI355X.

    python separate_dsd.py -i <inputfile> -o <outputdir> -m <path_to_model.pkl>

Same options, hard-coded hyper-parameters and output file names as the
reference's main() (examples/dsd100/separate_dsd.py); the work is done by
deepconvsep_amd (HIP kernels behind libdcs.so).  The reference's getopt long
option list contains the typo "--mfile" so only -m works there; here --mfile
works as well.
"""
import getopt
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

from deepconvsep_amd.separation import (generate_overlapadd, load_model, overlapadd, overlapadd_multi)  # noqa: E402,F401
from deepconvsep_amd.separation import train_auto as _train_auto  # noqa: E402
from deepconvsep_amd.transform import compute_file, compute_inverse  # noqa: E402,F401

USAGE = 'python separate_dsd.py -i <inputfile> -o <outputdir> -m <path_to_model.pkl>'


def train_auto(filein, outdir, model, scale_factor=0.3, time_context=30, overlap=20, batch_size=32, input_size=513,
               frameSize=None, hopSize=None):
    return _train_auto('dsd', filein, outdir, model, scale_factor, time_context, overlap, batch_size, input_size,
                       frameSize, hopSize)


def main(argv):
    try:
        opts, args = getopt.getopt(argv, "hi:o:m:", ["ifile=", "odir=", "mfile="])
    except getopt.GetoptError:
        print(USAGE)
        sys.exit(2)
    for opt, arg in opts:
        if opt == '-h':
            print(USAGE)
            sys.exit()
        elif opt in ("-i", "--ifile"):
            inputfile = arg
        elif opt in ("-o", "--odir"):
            outdir = arg
        elif opt in ("-m", "--mfile"):
            model = arg
    train_auto(inputfile, outdir, model, 0.3, 30, 25, 32, 513)


if __name__ == "__main__":
    main(sys.argv[1:])
