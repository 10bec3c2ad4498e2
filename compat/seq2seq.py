"""`from seq2seq import Seq2seq` (styletransformer/main.py:9) -> nast_amd.seq2seq."""
import nast_amd.utils as _U
from nast_amd.seq2seq import Seq2seq as _Seq2seq, item, reserve_streams  # noqa: F401


class Seq2seq(_Seq2seq):
    def __init__(self, param):
        # main.py:25 calls the `utils` it imported; when that is the reference's own package ours still needs the device
        _U.cuda_init(0, param.args.cuda)
        super().__init__(param)
