"""Byte-level framing helpers of the codec bitstreams.

`BytesListUtils` reproduces the framing of
/root/reference/lib/entropy_models/hyperprior/noisy_deep_factorized/utils.py:8-77 (a list of byte strings with
variable-width length fields) so that streams are interchangeable with the reference's; `int_to_bytes` /
`bytes_to_int` are the little-endian helpers of
/root/reference/models/convolutional/lossy_coord_lossy_color/geo_lossl_em.py:320-327.
"""
import io
import math
from typing import List, Optional


def int_to_bytes(x: int, length: int, byteorder: str = 'little', signed: bool = False) -> bytes:
    if not isinstance(x, int):
        raise TypeError('int expected')
    return x.to_bytes(length, byteorder=byteorder, signed=signed)


def bytes_to_int(s: bytes, byteorder: str = 'little', signed: bool = False) -> int:
    if not isinstance(s, bytes):
        raise TypeError('bytes expected')
    return int.from_bytes(s, byteorder=byteorder, signed=signed)


def _width_of(length: int) -> int:
    """bytes needed to store `length` (at least one)"""
    return max(1, (length.bit_length() + 7) // 8)


def _header_len(count: int, bits_per_item: int) -> int:
    # The reference sizes the header as ceil(count / items_per_byte + 0.25) bytes (utils.py:27): room for the marker bit
    # plus the width codes, with a spare byte for some counts.  Kept as is: it is part of the stream format.
    per_byte = 8 // bits_per_item
    return math.ceil(count / per_byte + 0.25)


class BytesListUtils:
    """Layout: header | length fields | payloads.
    header  = big-endian bit string '1' + per string (width - 1) on 1 bit (all widths <= 2) or 2 bits (widths <= 3;
              then bit 7 of the first header byte is additionally set);
    lengths = each string's length on `width` bytes, little-endian."""

    @staticmethod
    def concat_bytes_list(bytes_list: List[bytes], bs_io: Optional[io.BytesIO] = None) -> Optional[bytes]:
        if len(bytes_list) < 2:
            raise ValueError('at least two strings are expected')
        widths = [_width_of(len(b)) for b in bytes_list]
        if max(widths) > 3:
            raise ValueError('strings longer than 2^24 - 1 bytes cannot be framed')
        item_bits = 2 if max(widths) > 2 else 1
        field = 1
        for w in widths:
            field = (field << item_bits) | (w - 1)
        n_header = _header_len(len(bytes_list), item_bits)
        header = bytearray(field.to_bytes(n_header, 'big'))
        if item_bits == 2:
            header[0] |= 0x80
        sink = bs_io if bs_io is not None else io.BytesIO()
        sink.write(bytes(header))
        for b, w in zip(bytes_list, widths):
            sink.write(len(b).to_bytes(w, 'little'))
        for b in bytes_list:
            sink.write(b)
        return None if bs_io is not None else sink.getvalue()

    @staticmethod
    def split_bytes_list(concat_bytes: Optional[bytes], bytes_list_len: int,
                         bs_io: Optional[io.BytesIO] = None) -> List[bytes]:
        if (concat_bytes is None) == (bs_io is None):
            raise ValueError('give either the bytes or a stream')
        src = bs_io if bs_io is not None else io.BytesIO(concat_bytes)
        first = src.read(1)[0]
        item_bits = 2 if first & 0x80 else 1
        n_header = _header_len(bytes_list_len, item_bits)
        field = int.from_bytes(bytes([first & 0x7f]) + src.read(n_header - 1), 'big')
        # drop the marker bit: the remaining bytes_list_len * item_bits bits are the width codes, first string first
        total = bytes_list_len * item_bits
        field &= (1 << total) - 1
        widths = [((field >> (total - (i + 1) * item_bits)) & ((1 << item_bits) - 1)) + 1
                  for i in range(bytes_list_len)]
        lengths = [int.from_bytes(src.read(w), 'little') for w in widths]
        return [src.read(n) for n in lengths]
