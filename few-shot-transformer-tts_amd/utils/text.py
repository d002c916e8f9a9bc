"""Byte tokenizer: UTF-8 bytes with pad=0 / eos=1 / sos=2 (reference utils/text.py:3-19)."""
pad_id = 0
eos_id = 1
sos_id = 2


def text_to_byte_sequence(text, use_sos=True, use_eos=True):
    seq = list(text.encode("utf-8"))
    if use_sos:
        seq = [sos_id] + seq
    if use_eos:
        seq = seq + [eos_id]
    return seq


def language_vec_to_id(lv):
    for i, v in enumerate(lv):
        if v > 0:
            return i
    return -1
