"""Host mirror of Madarch.Lights (reference madarch/madarch-lights.ads:7-37): a
light KIND is a name plus its component list (PointLight, SpotLight)."""


class Light:
    def __init__(self, name, comps):
        self.name = name
        self.comps = list(comps)

    def __repr__(self):
        return "Light(%r)" % self.name


def Create(Name, Comps):
    return Light(Name, Comps)


def Get_Name(l):
    return l.name


def Get_Components(l):
    return list(l.comps)


from . import point_lights, spot_lights  # noqa: E402,F401
