#!/usr/bin/env python3
"""Drop-in for the reference's examples/bach10_scoreinformed/separate_bach10.py, running on MI355X.

    python separate_bach10.py -i <inputfile> -o <outputdir> -m <path_to_model.pkl>

The four score files ``bassoon_b.txt, clarinet_b.txt, saxophone_b.txt, violin_b.txt`` (lines ``onset,offset,note``)
are read from the directory of the input wav, as in the reference (:455, :516).  Same hard-coded hyper-parameters
(:572) and output names ``<name>_{bassoon,clarinet,saxphone,violin}.wav`` (:453, :545).  The reference script itself
does not run as shipped (``bisect``/``itertools``/``util``/``slicefft_slices`` are not imported, ``sources``,
``toverlap`` and ``source[i]`` are undefined or misspelt names); the behaviour reproduced here is the one its helper
code in ``util.py`` defines.
"""
import getopt
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

from deepconvsep_amd.score import expandMidi, filterSpec, getMidiNum, str2midi  # noqa: E402,F401
from deepconvsep_amd.separation import generate_overlapadd, load_model, overlapadd_multi  # noqa: E402,F401
from deepconvsep_amd.separation import train_auto as _train_auto  # noqa: E402
from deepconvsep_amd.transform import compute_file, compute_inverse  # noqa: E402,F401

USAGE = 'python separate_bach10.py -i <inputfile> -o <outputdir> -m <path_to_model.pkl>'


def train_auto(filein, outdir, model, scale_factor=0.3, time_context=30, overlap=20, batch_size=32, input_size=2049,
               frameSize=4096, hopSize=512):
    return _train_auto('bach10_si', filein, outdir, model, scale_factor, time_context, overlap, batch_size, input_size,
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
    train_auto(inputfile, outdir, model, 0.3, 30, 25, 32, 2049, 4096, 512)


if __name__ == "__main__":
    main(sys.argv[1:])
