"""Map: the static occupancy grid (reference: gym_collision_avoidance/envs/Map.py).

The reference rebuilds `self.map` (static grid + every agent as a disc) on the host each step and hands it to the
map-based sensors.  Here only the STATIC grid lives on the host; the per-step dynamic grid is assembled per env in
LDS by the scan kernel (csrc/cagpu_scan.inc).  `map_filename` is not supported (the reference's loader needs
imageio + scipy.misc.imresize); pass the grid as a bool array instead."""
import numpy as np


class Map(object):
    def __init__(self, x_width, y_width, grid_cell_size, map_filename=None, static_map=None):
        if map_filename is not None:
            raise NotImplementedError("loading map images is not supported: pass static_map=<bool array>")
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
