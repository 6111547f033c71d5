/* TEST INFRASTRUCTURE ONLY — plain-data scene description used by both oracle builds.
 * Field meanings follow the reference classes: Sphere(center, radius) main.cpp:198-227,
 * Plane(center, normal) main.cpp:171-195, Material/Chrome/Glass main.cpp:230-306,
 * Camera main.cpp:132-158. */
#ifndef PTOR_TYPES_H
#define PTOR_TYPES_H
#include <stdint.h>

enum { PTOR_SPHERE = 0, PTOR_PLANE = 1, PTOR_RECTANGLE = 2 /* Rectangle(p1,p2,p3,p4) raytrace3.cpp:244-276 */,
       PTOR_RECTANGLE_MT = 3 /* Rectangle(p1,p2,p3,p4) light_tracing.cpp:477-575 */ };
enum { PTOR_DIFFUSE = 0, PTOR_CHROME = 1, PTOR_GLASS = 2, PTOR_MEASURED = 3 };
enum { PTOR_EST_PT = 0, PTOR_EST_NEE = 1, PTOR_EST_BIDIR = 2 /* bidirectional.cpp:374-497 */ };

typedef struct ptor_body {
    int32_t shape;
    int32_t material;
    double center[3];
    double normal[3]; /* planes: the value STORED by the Plane ctor (already normalised) */
    double radius;
    double color[3];
    double emission[3];
    double ior;
    double reflection;
    int32_t brdf_id;
    int32_t pad;
    double corners[12]; /* rectangles: p1, p2, p3, p4 as passed to the ctor */
} ptor_body;

typedef struct ptor_camera {
    double origin[3];
    double topleft[3];
    double topright[3];
    double bottomleft[3];
} ptor_camera;

#endif
