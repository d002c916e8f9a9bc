// Dropout sites of the model path: ONE table for the training segments (engine.hip), the autoregressive loop (decode.hip) and the read-only
// C-ABI query b2s_dropout_site (include/b2s_hip.h), so that a checker never restates an op id by hand.
// Reference: every F.dropout call of transformer/modules.py:18,55,64,67,120,132,138,141, transformer/attention.py:89 and transformer/tacotron.py:58,62,89.
//
// The mask of a site is keep(idx) = hash32(idx * 0x9E3779B1 + key) >= p * 2^32 (b2s_common.h) with key = make_drop(p, seed, op id): `seed` is the
// seed argument of the segment call (b2s_encoder_forward / b2s_decoder_forward / b2s_postnet_forward / b2s_decode_begin) and
//   training segments:  op id = segment * 4096 + layer * 32 + k
//   decode loop:        op id = decode_base + layer, and the key is additionally salted with the frame index t (salt rule below)
// idx is the flat row-major index of the element in the tensor the site acts on:
//   B2S_DROP_ROWS  activations [rows, C] (token-major: row = b * L + position; decode loop: row = b):  idx = row * C + column
//   B2S_DROP_ATTN  softmax weights [B, H, Lq, Lk]:  training: the row-seed / key-quad rule of b2s_common.h (b2s_keep_w) with row = (b * H + h) * Lq + q --
//                  seed = hash32(row * 0x9E3779B1 + key), y = x ^ (x >> 16) with x = seed + (k >> 2) * 0x9E3779B1, word = y * ((k & 2) ? 0xC2B2AE35 :
//                  0x85EBCA6B), key k uses the (k & 1)-th 16-bit field of its word and is kept when (int16) field >= ((p * 2^32) >> 16) - 32768;
//                  decode loop: keep(idx) as above with idx = (b * H + h) * 4096 + k
// Frame salts of the decode loop (key ^= hash32(...)):
//   B2S_SALT_ROWS   t * 2246822519 + 3266489917      (prenet, residual and FFN-hidden sites)
//   B2S_SALT_EMBED  t + 0x9e3779b9                   (the position-encoding dropout)
//   B2S_SALT_ATTN   t * 2654435761 + 77              (attention weights)
#pragma once
#include <stdint.h>
#include <string.h>

enum DropSite {
    DS_ENC_EMBED = 0, DS_ENC_ATTN, DS_ENC_ATTN_RES, DS_ENC_FFN_HID, DS_ENC_FFN_RES,
    DS_DEC_PRENET0, DS_DEC_PRENET1, DS_DEC_EMBED, DS_DEC_SELF_ATTN, DS_DEC_SELF_RES, DS_DEC_CROSS_ATTN, DS_DEC_CROSS_RES, DS_DEC_FFN_HID, DS_DEC_FFN_RES,
    DS_POST_CONV, DS_COUNT
};
enum { B2S_DROP_ROWS = 0, B2S_DROP_ATTN = 1 };
enum { B2S_SALT_NONE = 0, B2S_SALT_ROWS = 1, B2S_SALT_EMBED = 2, B2S_SALT_ATTN = 3 };

struct DropSiteInfo {
    const char* name;     // what b2s_dropout_site is asked for
    int seg, k;           // training op id = seg * 4096 + layer * 32 + k
    int decode_base;      // decode-loop op id = decode_base + layer (0: the site does not exist in the loop)
    int kind;             // B2S_DROP_ROWS / B2S_DROP_ATTN
    int salt;             // decode loop: B2S_SALT_*
};
static const DropSiteInfo g_drop_sites[DS_COUNT] = {
    {"encoder.embed",        1, 1, 0,    B2S_DROP_ROWS, B2S_SALT_NONE},       // modules.py:55   dropout(x + pe * pe_scale)
    {"encoder.attn",         1, 2, 0,    B2S_DROP_ATTN, B2S_SALT_NONE},       // attention.py:89 dropout(softmax weights)
    {"encoder.attn_res",     1, 3, 0,    B2S_DROP_ROWS, B2S_SALT_NONE},       // modules.py:64   x + dropout(attention output)
    {"encoder.ffn_hidden",   1, 4, 0,    B2S_DROP_ROWS, B2S_SALT_NONE},       // modules.py:18   dropout(relu(input_layer))
    {"encoder.ffn_res",      1, 5, 0,    B2S_DROP_ROWS, B2S_SALT_NONE},       // modules.py:67   x + dropout(ffn output)
    {"decoder.prenet0",      2, 1, 9001, B2S_DROP_ROWS, B2S_SALT_ROWS},       // tacotron.py:58
    {"decoder.prenet1",      2, 2, 9002, B2S_DROP_ROWS, B2S_SALT_ROWS},       // tacotron.py:62
    {"decoder.embed",        2, 3, 9003, B2S_DROP_ROWS, B2S_SALT_EMBED},      // modules.py:120  dropout(shifted targets + pe * pe_scale)
    {"decoder.self_attn",    2, 4, 9010, B2S_DROP_ATTN, B2S_SALT_ATTN},       // attention.py:89
    {"decoder.self_res",     2, 5, 9020, B2S_DROP_ROWS, B2S_SALT_ROWS},       // modules.py:132
    {"decoder.cross_attn",   2, 6, 9030, B2S_DROP_ATTN, B2S_SALT_ATTN},       // attention.py:89
    {"decoder.cross_res",    2, 7, 9040, B2S_DROP_ROWS, B2S_SALT_ROWS},       // modules.py:138
    {"decoder.ffn_hidden",   2, 8, 9050, B2S_DROP_ROWS, B2S_SALT_ROWS},       // modules.py:18
    {"decoder.ffn_res",      2, 9, 9060, B2S_DROP_ROWS, B2S_SALT_ROWS},       // modules.py:141
    {"postnet.conv",         3, 1, 0,    B2S_DROP_ROWS, B2S_SALT_NONE},       // tacotron.py:89  dropout after every conv + batchnorm (+ tanh)
};
inline uint32_t drop_op(DropSite s, int layer) { return (uint32_t)(g_drop_sites[s].seg * 4096 + layer * 32 + g_drop_sites[s].k); }
inline uint32_t drop_op_decode(DropSite s, int layer) { return (uint32_t)(g_drop_sites[s].decode_base + layer); }
