"""Piece scores straight from a serialized ModelProto (src/sentencepiece_model.proto: ``repeated SentencePiece pieces = 1``
with ``optional float score = 2``, default 0) -- what ``GetScore`` returns (src/sentencepiece_processor.cc ``GetScore``:
``model_->GetScore(id)``).  Host-side accessor only; the device tables are compiled by csrc/tables.cc."""
import struct


def _varint(b, i):
    v = s = 0
    while True:
        c = b[i]
        i += 1
        v |= (c & 0x7F) << s
        s += 7
        if not c & 0x80:
            return v, i


def _skip(b, i, wt):
    if wt == 0:
        return _varint(b, i)[1]
    if wt == 1:
        return i + 8
    if wt == 2:
        ln, i = _varint(b, i)
        return i + ln
    if wt == 5:
        return i + 4
    raise ValueError("wire type %d" % wt)


def scores(blob):
    out = []
    i, n = 0, len(blob)
    while i < n:
        tag, i = _varint(blob, i)
        num, wt = tag >> 3, tag & 7
        if num == 1 and wt == 2:
            ln, i = _varint(blob, i)
            j, end, sc = i, i + ln, 0.0
            while j < end:
                t2, j = _varint(blob, j)
                if t2 >> 3 == 2 and t2 & 7 == 5:
                    sc = struct.unpack_from("<f", blob, j)[0]
                    j += 4
                else:
                    j = _skip(blob, j, t2 & 7)
            out.append(sc)
            i = end
        else:
            i = _skip(blob, i, wt)
    return out
