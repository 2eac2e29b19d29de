"""Host mirror of Madarch.Values (reference madarch/madarch-values.ads:8-27):
a tagged value is Vector3 | Float | Int, fp32 / int32."""
import numpy as np

Vector3_Kind, Float_Kind, Int_Kind = 0, 1, 2  # = Value_Kind order, values.ads:8


class Value:
    __slots__ = ("kind", "data")

    def __init__(self, kind, data):
        self.kind = kind
        self.data = data

    def __repr__(self):
        return "Value(%s, %r)" % (("Vector3", "Float", "Int")[self.kind], self.data)


def Vector3(x):  # values.ads:25
    v = np.asarray(x, dtype=np.float32).reshape(3)
    return Value(Vector3_Kind, v.copy())


def Float(x):  # values.ads:26
    return Value(Float_Kind, np.float32(x))


def Int(x):  # values.ads:27
    return Value(Int_Kind, np.int32(x))
