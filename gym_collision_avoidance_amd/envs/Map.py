"""Map: the static occupancy grid (reference: gym_collision_avoidance/envs/Map.py).

The reference rebuilds `self.map` (static grid + every agent as a disc) on the host each step and hands it to the
map-based sensors.  Here only the STATIC grid lives on the host; the per-step dynamic grid is assembled per env in
LDS by the scan kernel (csrc/cagpu_scan.inc).  `map_filename` is read like the reference does (Map.py:14-24: image ->
nearest-neighbour resize to the grid -> np.invert -> bool, i.e. dark pixels are obstacles) with whichever of imageio /
PIL is installed; `static_map` passes the grid directly as a bool array."""
import numpy as np


def load_map_image(map_filename, dims):
    """Map.py:17-22 without scipy.misc.imresize (removed from SciPy): the image as grey levels, imresize's bytescale
    (min .. max stretched to 0 .. 255 when the image has to be resized -- for NON-uint8 pixel data only: scipy's bytescale
    returns uint8 input unchanged, so an 8-bit map whose brightest pixel is below 255 keeps its levels and every pixel
    != 255 is an obstacle, exactly as in the reference), nearest-neighbour resize, invert: dark pixels are obstacles."""
    img = None
    try:
        from PIL import Image
        img = np.asarray(Image.open(map_filename).convert("L"))   # palette / RGB(A) / 1-bit -> grey levels 0 .. 255
    except ImportError:
        Image = None
        try:
            import imageio
            img = np.asarray(imageio.imread(map_filename))
        except ImportError:
            raise ImportError("reading a map image needs Pillow or imageio (neither is installed); pass the occupancy grid "
                              "itself with Map(..., static_map=<bool array>) / env.set_static_map(<bool array>)")
        if img.ndim == 3:
            img = img[..., :3].mean(axis=-1).astype(img.dtype) if img.dtype == np.uint8 else img[..., :3].mean(axis=-1)
        if img.dtype == bool:
            img = img.astype(np.uint8) * 255
    img = np.asarray(img)
    if img.shape != tuple(dims):
        if Image is None:
            raise ImportError("resizing a %s map image to %s needs Pillow" % (img.shape, tuple(dims)))
        if img.dtype == np.uint8:                               # scipy.misc.bytescale: uint8 data is returned as it is
            scaled = img
        else:                                                   # ... anything else is stretched min .. max -> 0 .. 255
            lo, hi = float(img.min()), float(img.max())
            scaled = np.zeros(img.shape, np.uint8) if hi == lo else np.clip((img - lo) * (255.0 / (hi - lo)) + 0.5, 0, 255).astype(np.uint8)
        img = np.asarray(Image.fromarray(scaled).resize((dims[1], dims[0]), Image.NEAREST))
    return np.invert(img.astype(np.uint8)).astype(bool)


class Map(object):
    def __init__(self, x_width, y_width, grid_cell_size, map_filename=None, static_map=None):
        if map_filename is not None:
            static_map = load_map_image(map_filename, (int(x_width / grid_cell_size), int(y_width / grid_cell_size)))
        self.x_width, self.y_width, self.grid_cell_size = x_width, y_width, grid_cell_size
        dims = (int(self.x_width / self.grid_cell_size), int(self.y_width / self.grid_cell_size))
        if static_map is None:
            self.static_map = np.zeros(dims, dtype=bool)
        else:
            self.static_map = np.asarray(static_map).astype(bool)
            assert self.static_map.shape == dims, (self.static_map.shape, dims)
        self.origin_coords = np.array([(self.x_width / 2.) / self.grid_cell_size,
                                       (self.y_width / 2.) / self.grid_cell_size])
        self.map = self.static_map  # the dynamic grid is never materialised on the host

    def world_coordinates_to_map_indices(self, pos):
        gx = int(np.floor(self.origin_coords[0] - pos[1] / self.grid_cell_size))
        gy = int(np.floor(self.origin_coords[1] + pos[0] / self.grid_cell_size))
        in_map = 0 <= gx < self.static_map.shape[0] and 0 <= gy < self.static_map.shape[1]
        return np.array([gx, gy]), in_map
