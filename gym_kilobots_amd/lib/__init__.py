from .kilobot import Kilobot, MotorKilobot, PhototaxisKilobot, SimplePhototaxisKilobot, SimpleVelocityControlKilobot,\
    SimpleAccelerationControlKilobot
from .body import Body, Circle, Quad, CornerQuad, Polygon, Triangle, LForm, TForm, CForm, World
from .light import Light, SinglePositionLight, CircularGradientLight, GradientLight, MomentumLight, CompositeLight, SmoothGridLight
