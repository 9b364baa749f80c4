import ctypes, sys
L = ctypes.CDLL("tests/_lds_poison.so"); L.lds_poison.argtypes = [ctypes.c_uint, ctypes.c_int]
print("poison rc", L.lds_poison(int(sys.argv[1], 16), 160 * 1024))
