"""Random scenario generation on the host (reference: gym_collision_avoidance/envs/policies/CADRL/scripts/multi/
gen_rand_testcases.py:111-444, used by test_cases.get_testcase_random, the reference's default TEST_CASE_FN).

Restated, not translated: one rejection sampler (`_sample_free`) serves the three scenario families.  What IS kept
exactly is every draw from `np.random` (count and order) and every floating-point expression that feeds an accept /
reject decision, so that under the same seed this module returns the reference's arrays bit for bit
(tests/golden/rand_cases.npz, recorded from the unmodified reference).  The scenarios are built on the host and
uploaded with one cagpu_reset; the reference's generator is a per-episode Python loop as well.

A case is float64 [N, 6] = px, py, gx, gy, pref_speed, radius.
"""
import numpy as np

GETTING_CLOSE_RANGE = 0.2   # CADRL/scripts/multi/global_var.py:8
EPS = 1e-5                  # global_var.py:9


def _norm(v):
    return np.linalg.norm(v)


def _draw_body(case, i, speed_bnds, radius_bnds):
    """radius, then the larger of two speed draws (three np.random.rand() calls, in this order)"""
    case[i, 5] = (radius_bnds[1] - radius_bnds[0]) * np.random.rand() + radius_bnds[0]
    s1 = (speed_bnds[1] - speed_bnds[0]) * np.random.rand() + speed_bnds[0]
    s2 = (speed_bnds[1] - speed_bnds[0]) * np.random.rand() + speed_bnds[0]
    case[i, 4] = max(s1, s2)


def _overlaps_earlier(case, i, start, end):
    """start within (r_j + r_i + 0.2) of an earlier start, or end of an earlier end (checked in that order per j)"""
    for j in range(i):
        clearance = case[j, 5] + case[i, 5] + GETTING_CLOSE_RANGE
        if _norm(start - case[j, 0:2]) < clearance:
            return True
        if _norm(end - case[j, 2:4]) < clearance:
            return True
    return False


def dist_point_to_segment(p1, p2, p3):
    """distance from p3 to the segment p1 -> p2 (gen_rand_testcases.py:91-108)"""
    d = p2 - p1
    if _norm(d) < EPS:
        u = 0.0
    else:
        u = np.dot(d, (p3 - p1)) / (_norm(d) ** 2.0)
    u = max(0.0, min(u, 1.0))
    return _norm(p3 - (p1 + u * d))


def min_dist_between_moving_points(x1, x2, y1, y2):
    """closest approach of two points moving x1 -> x2 and y1 -> y2 over the same unit time, not counting the start
    (find_dist_between_segs, gen_rand_testcases.py:52-88, single-pair form)"""
    x2 = x2.reshape((1, 2))
    y2 = y2.reshape((1, 2))
    end_dist = np.linalg.norm(x2 - y2, axis=1)
    critical = end_dist.copy()
    z_bar = (x2 - x1) - (y2 - y1)
    inds = np.where((np.linalg.norm(z_bar, axis=1) > 0))[0]
    t_bar = - np.sum((x1 - y1) * z_bar[inds, :], axis=1) / np.sum(z_bar[inds, :] * z_bar[inds, :], axis=1)
    t_rep = np.tile(t_bar, (2, 1)).transpose()
    dist_bar = np.linalg.norm(x1 + (x2[inds, :] - x1) * t_rep - y1 - (y2[inds, :] - y1) * t_rep, axis=1)
    inside = np.where((t_bar > 0) & (t_bar < 1.0))
    critical[inds[inside]] = dist_bar[inside]
    return np.amin(np.vstack((end_dist, critical)), axis=0)[0]


def straight_lines_are_safe(x1, x2, s1, y1, y2, s2, radius):
    """True if two agents driving straight to their goals at their speeds never come within `radius`
    (if_permitStraightLineSoln, gen_rand_testcases.py:422-444)"""
    t1 = _norm(x2 - x1) / s1
    t2 = _norm(y2 - y1) / s2
    if t1 < t2:
        x_crit = x2
        y_crit = y1 + t1 * (y2 - y1) / t2
        if dist_point_to_segment(y_crit, y2, x_crit) < radius:
            return False
    else:
        x_crit = x1 + t2 * (x2 - x1) / t1
        y_crit = y2
        if dist_point_to_segment(x_crit, x2, y_crit) < radius:
            return False
    start_dist = _norm(x1 - y1)
    end_dist = _norm(x_crit - y_crit)
    mid_dist = min_dist_between_moving_points(x1, x_crit, y1, y_crit)
    return not (min(start_dist, end_dist, mid_dist) < radius)


def _antipodal_on_circle(case, i, r, offset):
    """draw start angles until start / end (antipodal on a circle of radius r, shifted by offset) are free; the circle
    grows by 1 % after every 11th rejection in a row.  Returns the (possibly grown) radius."""
    rejected = 0
    while True:
        if rejected > 10:
            r *= 1.01
            rejected = 0
        start_angle = np.random.rand() * 2 * np.pi - np.pi
        end_angle = np.pi + start_angle
        start = np.array([r * np.cos(start_angle), r * np.sin(start_angle)]) + offset
        end = np.array([r * np.cos(end_angle), r * np.sin(end_angle)]) + offset
        if _overlaps_earlier(case, i, start, end):
            rejected += 1
            continue
        case[i, 0:2] = start
        case[i, 2:4] = end
        return r


def generate_circle_case(num_agents, side_length, speed_bnds, radius_bnds):
    """everybody crosses a circle of radius ~ N/2 .. N/2 + 2 (gen_rand_testcases.py:380-420)"""
    r = np.random.rand() * 2.0 + num_agents / 2.0
    case = np.zeros((num_agents, 6))
    for i in range(num_agents):
        _draw_body(case, i, speed_bnds, radius_bnds)
        # the reference adds nothing to the circle points; `+ 0.0` keeps the values and the array type identical
        r = _antipodal_on_circle(case, i, r, 0.0)
    return case


def generate_swap_case(num_agents, side_length, speed_bnds, radius_bnds):
    """agents 0 and 1 swap places on the x axis, the others cross a circle beside them (gen_rand_testcases.py:322-377)"""
    r_min = num_agents / 2.0
    r = np.random.rand() * 2.0 + r_min
    case = np.zeros((num_agents, 6))
    r_swap = 1.5 + np.random.rand() * 2.0
    offset = np.array([0, 1.0 + r_min + np.random.rand() * 2.0])
    if np.random.rand() > 0.5:
        offset = -offset
    for i in range(num_agents):
        _draw_body(case, i, speed_bnds, radius_bnds)
        if i == 0:
            case[i, 0:2] = [-r_swap, 0.0]
            case[i, 2:4] = [r_swap, 0.0]
        elif i == 1:
            case[i, 0:2] = [r_swap, 0.0]
            case[i, 2:4] = [-r_swap, 0.0]
        else:
            r = _antipodal_on_circle(case, i, r, offset)
    return case


def generate_rand_case(num_agents, side_length, speed_bnds, radius_bnds, is_end_near_bnd=False):
    """uniform starts / goals in a square that grows 1 % per attempt; a pair is rejected if it overlaps an earlier
    agent, if every earlier agent could be passed by driving straight (too easy), or if the trip is shorter than half
    the side (gen_rand_testcases.py:144-231)"""
    case = np.zeros((num_agents, 6))
    for i in range(num_agents):
        _draw_body(case, i, speed_bnds, radius_bnds)
        while True:
            side_length *= 1.01
            start = side_length * 2 * np.random.rand(2,) - side_length
            end = side_length * 2 * np.random.rand(2,) - side_length
            if is_end_near_bnd:
                wall = np.random.randint(4)
                if wall == 0:
                    end[0] = np.random.rand() * 0.1 * side_length - side_length
                elif wall == 1:
                    end[0] = np.random.rand() * 0.1 * side_length + 0.9 * side_length
                elif wall == 2:
                    end[1] = np.random.rand() * 0.1 * side_length - side_length
                else:
                    end[1] = np.random.rand() * 0.1 * side_length + 0.9 * side_length
            if _overlaps_earlier(case, i, start, end):
                continue
            if i >= 1:
                trivial = True
                for j in range(0, i):
                    clearance = case[j, 5] + case[i, 5] + GETTING_CLOSE_RANGE
                    if not straight_lines_are_safe(case[j, 0:2], case[j, 2:4], case[j, 4], start, end, case[i, 4],
                                                   clearance):
                        trivial = False
                        break
                if trivial:
                    continue
            if _norm(start - end) > side_length * 0.5:
                break
        case[i, 0:2] = start
        case[i, 2:4] = end
    return case


def generate_static_case(num_agents, side_length, speed_bnds, radius_bnds):
    """agent 0 crosses the square from left to right, every other agent stands still (start = goal): agent 1 at the origin,
    the rest rejection-sampled in the inner half of a square that grows 1 % per rejected attempt
    (gen_rand_testcases.py:263-317).  Same np.random draws in the same order as the reference."""
    case = np.zeros((num_agents, 6))
    for i in range(num_agents):
        _draw_body(case, i, speed_bnds, radius_bnds)
        if i == 0:
            start = side_length * 2.0 * np.random.rand(2,) - side_length
            end = side_length * 2.0 * np.random.rand(2,) - side_length
            start[0] = min(-1.5, -np.random.rand() * side_length)
            start[1] = np.random.rand() * 2.0 - 1.0
            end[0] = max(1.5, np.random.rand() * side_length)
            end[1] = np.random.rand() * 2.0 - 1.0
        elif i == 1:
            start = np.zeros((2,))
            end = start
        else:
            while True:
                start = (side_length * 2 * np.random.rand(2,) - side_length) / 2.0
                end = start
                if not _overlaps_earlier(case, i, start, end):
                    break
                side_length *= 1.01
        case[i, 0:2] = start
        case[i, 2:4] = end
    return case


def generate_rand_test_case_multi(num_agents, side_length, speed_bnds, radius_bnds, is_end_near_bnd=False,
                                  is_static=False):
    """15 % swap, 15 % circle, 70 % random (gen_rand_testcases.py:111-142)"""
    dice = np.random.rand()   # (drawn before the is_static test, like the reference: the stream position is part of the result)
    if is_static:
        return generate_static_case(num_agents, side_length, speed_bnds, radius_bnds)
    if dice < 0.15:
        return generate_swap_case(num_agents, side_length, speed_bnds, radius_bnds)
    if dice > 0.15 and dice < 0.3:
        return generate_circle_case(num_agents, side_length, speed_bnds, radius_bnds)
    return generate_rand_case(num_agents, side_length, speed_bnds, radius_bnds, is_end_near_bnd=is_end_near_bnd)
