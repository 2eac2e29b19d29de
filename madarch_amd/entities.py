"""Host mirror of Madarch.Entities (reference madarch/madarch-entities.adb:2-43):
an entity is an ordered bag of (component, value), searched linearly."""


class Entity:
    def __init__(self, values):
        self.values = [[c, v] for (c, v) in values]

    def Get(self, comp):  # entities.adb:9-20
        for c, v in self.values:
            if c is comp:
                return v
        raise RuntimeError("Entity does not have given component.")

    def Set(self, comp, value):  # entities.adb:22-34
        for cv in self.values:
            if cv[0] is comp:
                cv[1] = value
                return
        raise RuntimeError("Entity does not have given component.")

    def Foreach(self, proc):  # entities.adb:36-43
        for c, v in self.values:
            proc(c, v)


def Create(values):
    return Entity(values)
