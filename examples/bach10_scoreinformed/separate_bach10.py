#!/usr/bin/env python3
"""Drop-in for the reference's examples/bach10_scoreinformed/separate_bach10.py, running on MI355X.

    python separate_bach10.py -i <inputfile> -o <outputdir> -m <path_to_model.pkl>

The four score files ``bassoon_b.txt, clarinet_b.txt, saxophone_b.txt, violin_b.txt`` (lines ``onset,offset,note``)
are read from the directory of the input wav, as in the reference (:455, :516).  Same hard-coded hyper-parameters
(:572) and output names ``<name>_{bassoon,clarinet,saxphone,violin}.wav`` (:453, :545).  The reference script itself
does not run as shipped (``bisect``/``itertools``/``util``/``slicefft_slices`` are not imported, ``sources``,
``toverlap`` and ``source[i]`` are undefined or misspelt names); the behaviour reproduced here is the one its helper
code in ``util.py`` defines.  Either .pkl of the reference loads: the 17-array graph (:388-447, trainCNNrwc.py) or the
single-branch 11-array graph of trainCNNrwc_samp.py:195-235.

One extra long option, ``--trainer-semantics``: harmonic masks divided by their sum over the instruments
(``LargeDatasetMask2.filterSpec``, dataset.py:862) and soft masks applied to the sum of the four input channels
(trainCNNrwc.py:258-263) -- what the TRAINER's own separation block computes and what its models were trained on; without it
the script's semantics (own maximum, :195; input channel 0, :485).
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
               frameSize=4096, hopSize=512, trainer_semantics=False):
    sem = ('sum', 'sum') if trainer_semantics else ('max', 'ch0')
    return _train_auto('bach10_si', filein, outdir, model, scale_factor, time_context, overlap, batch_size, input_size,
                       frameSize, hopSize, score_normalise=sem[0], score_mixture=sem[1])


def main(argv):
    try:
        opts, args = getopt.getopt(argv, "hi:o:m:", ["ifile=", "odir=", "mfile=", "trainer-semantics"])
    except getopt.GetoptError:
        print(USAGE)
        sys.exit(2)
    trainer = False
    for opt, arg in opts:
        if opt == '--trainer-semantics':
            trainer = True
        elif opt == '-h':
            print(USAGE)
            sys.exit()
        elif opt in ("-i", "--ifile"):
            inputfile = arg
        elif opt in ("-o", "--odir"):
            outdir = arg
        elif opt in ("-m", "--mfile"):
            model = arg
    train_auto(inputfile, outdir, model, 0.3, 30, 25, 32, 2049, 4096, 512, trainer_semantics=trainer)


if __name__ == "__main__":
    main(sys.argv[1:])
