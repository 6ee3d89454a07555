"""Generate tests/golden/wire_vectors.json: the serialized bytes of the reference's KV records
(src/image_search.proto:3-27 -- what PilafProxy / MemcachedProxy / RedisProxy put on the wire with
SerializeToString, pilaf_proxy.h:39-62), produced by a REAL protobuf runtime.

The image has no protoc, so the message descriptors are assembled here field by field from the .proto's
declarations (names, numbers, types, labels as in image_search.proto:3-27); the installed `google.protobuf`
runtime does every byte of the encoding.  Run in the authoring container:
    python tests/golden/make_wire_vectors.py
The output is data only (field values in, wire bytes out); tests/test_oracle_cpu.py feeds the same field values
to verticut_amd/host/verticut_wire.hpp (tests/cpp/wire_test.cc) on any box and compares byte for byte.
"""
import json
import os

import numpy as np
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
import google.protobuf

F = descriptor_pb2.FieldDescriptorProto
fd = descriptor_pb2.FileDescriptorProto()
fd.name = "image_search.proto"
fd.syntax = "proto2"


def message(name, fields):
    m = fd.message_type.add()
    m.name = name
    for fname, number, ftype, label, type_name in fields:
        f = m.field.add()
        f.name, f.number, f.type, f.label = fname, number, ftype, label
        if type_name:
            f.type_name = type_name


# image_search.proto:3-27, declaration by declaration
message("ID", [("id", 1, F.TYPE_UINT32, F.LABEL_REQUIRED, None)])                                   # :3-5
message("BinaryCode", [("code", 1, F.TYPE_BYTES, F.LABEL_REQUIRED, None)])                           # :7-9
message("HashIndex", [("table_id", 1, F.TYPE_UINT32, F.LABEL_REQUIRED, None),                        # :11-14
                      ("index", 2, F.TYPE_UINT32, F.LABEL_REQUIRED, None)])
message("ID_Code_Pair", [("id", 1, F.TYPE_UINT32, F.LABEL_REQUIRED, None),                           # :16-19
                         ("code", 2, F.TYPE_BYTES, F.LABEL_REQUIRED, None)])
message("ImageList", [("images", 1, F.TYPE_UINT32, F.LABEL_REPEATED, None)])                         # :21-23 (unused)
message("Image_List", [("images", 1, F.TYPE_MESSAGE, F.LABEL_REPEATED, ".ID_Code_Pair")])            # :25-27

pool = descriptor_pool.DescriptorPool()
pool.Add(fd)
cls = {n: message_factory.GetMessageClass(pool.FindMessageTypeByName(n)) for n in ("ID", "BinaryCode", "HashIndex", "ID_Code_Pair", "Image_List")}

rng = np.random.default_rng(20131012)
edge_u32 = [0, 1, 127, 128, 300, 16383, 16384, 2097151, 2097152, 268435455, 268435456, 0x7FFFFFFF, 0x80000000, 0xFFFF8001, 0xFFFFFFFF]
out = {"source": "google.protobuf %s runtime, descriptors assembled from image_search.proto:3-27" % google.protobuf.__version__,
       "id": [], "binarycode": [], "hashindex": [], "imagelist": []}

for v in edge_u32 + [int(x) for x in rng.integers(0, 1 << 32, 10)]:
    m = cls["ID"]()
    m.id = v
    out["id"].append({"id": v, "wire": m.SerializeToString().hex()})

codes = [b"", b"0123456789123456", bytes(8), bytes([255] * 32), bytes(range(127, 127 + 64)), bytes(200)]   # 200: 2-byte length varint
codes += [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in (8, 16, 16, 32, 64)]
for c in codes:
    m = cls["BinaryCode"]()
    m.code = c
    out["binarycode"].append({"code": c.hex(), "wire": m.SerializeToString().hex()})

for t, i in [(0, 0), (3, 0xFFFF8001), (1, 127), (2, 128), (7, 0xFFFFFFFF), (255, 65535), (4, 0x80000000)] + \
        [(int(rng.integers(0, 8)), int(rng.integers(0, 1 << 32))) for _ in range(10)]:
    m = cls["HashIndex"]()
    m.table_id, m.index = t, i
    out["hashindex"].append({"table_id": t, "index": i, "wire": m.SerializeToString().hex()})

lists = [[], [(0, b"0123456789123456")], [(i * 1000000, b"0123456789123456") for i in range(3)]]
for n_entries, nbytes in ((1, 8), (5, 16), (40, 16), (7, 32), (3, 64), (2, 200)):
    lists.append([(int(rng.integers(0, 1 << 32)), rng.integers(0, 256, nbytes, dtype=np.uint8).tobytes()) for _ in range(n_entries)])
lists.append([(v, b"\x00" * 16) for v in edge_u32])
for entries in lists:
    m = cls["Image_List"]()
    for i, c in entries:
        p = m.images.add()
        p.id, p.code = i, c
    out["imagelist"].append({"images": [[i, c.hex()] for i, c in entries], "wire": m.SerializeToString().hex()})

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wire_vectors.json")
with open(path, "w") as f:
    json.dump(out, f, separators=(",", ":"))
print(path, {k: len(v) for k, v in out.items() if isinstance(v, list)}, os.path.getsize(path))
