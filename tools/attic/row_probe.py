import ctypes, numpy as np, os
lib = ctypes.CDLL(os.path.join(os.getcwd(), "build", "row_proto.so"))
o = np.zeros(256, np.uint32)
lib.row_proto_swap_probe(o.ctypes.data_as(ctypes.c_void_p))
for name, k in (("p16 r0", 0), ("p16 r1", 64), ("p32 r0", 128), ("p32 r1", 192)):
    print(name, [int(o[k + 16 * r]) for r in range(4)], "(first lane of each row; a = lane, b = 100 + lane)")
