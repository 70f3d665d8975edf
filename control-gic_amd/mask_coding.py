"""BinaryCoding -- drop-in for CGIC/tools/mask_coding.py:8-96 (1 bit per mask
element, MSB first, same 8-bit pad header as the Huffman streams)."""
from .indices_coding import _StreamCoder, _Table


class BinaryCoding(_StreamCoder):
    def __init__(self):
        self._table = _Table.binary()
        self.codes = {0: "0", 1: "1"}
        self.reverse_mapping = {"0": 0, "1": 1}

    @property
    def table(self):
        return self._table
