"""Headless stand-in for Madarch.Windows (reference madarch/madarch-windows.ads:12-31).
There is no display on an MI355X box: a window is just a frame size, and the
frame is read back with Renderers.Read_Framebuffer instead of Swap_Buffers."""


class Window:
    def __init__(self, width, height, title):
        self.Width, self.Height, self.Title = int(width), int(height), title
        self._opened = True

    def Is_Opened(self):
        return self._opened

    def Close(self):
        self._opened = False

    def Poll_Events(self):
        pass


def Open(Width, Height, Title=""):
    return Window(Width, Height, Title)
